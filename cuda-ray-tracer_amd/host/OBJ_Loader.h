// Host mirror of the reference's OBJ/MTL loader API (OBJ_Loader.h:16-325): objl::Loader with LoadFile and the
// LoadedMeshes / LoadedVertices / LoadedIndices / LoadedMaterials results that Model's constructor consumes.
//
// Written from scratch as a single-pass parser; what it reproduces of the reference loader is the OUTPUT
// contract, because that fixes vertex and triangle order and therefore which triangle wins an exact tie:
//   - every face corner becomes a new vertex (no sharing), in file order; indices are global into LoadedVertices
//     (OBJ_Loader.cpp:329-357);
//   - a mesh ends at each o/g line and at each usemtl that follows faces (OBJ_Loader.cpp:238-289, 359-390); the i-th
//     usemtl names the material of the i-th mesh (OBJ_Loader.cpp:441-455);
//   - corners without a normal get the face's cross(v0 - v1, v2 - v1) (OBJ_Loader.cpp:562-573);
//   - triangles keep their corner order; a quad (0,1,2,3) becomes (0,1,3), (1,2,3); larger polygons are ear-clipped
//     in the reference's order (OBJ_Loader.cpp:576-741);
//   - negative (relative) indices count from the end (OBJ_Loader.h:269-278);
//   - Mesh objects carry their vertices but no indices (the reference's Mesh constructor drops them,
//     OBJ_Loader.cpp:91-94).
#pragma once
#include <string>
#include <vector>

namespace objl {

struct Vector2 {
    float X = 0.0f, Y = 0.0f;
    Vector2() {}
    Vector2(float x, float y) : X(x), Y(y) {}
    bool operator==(const Vector2& o) const { return X == o.X && Y == o.Y; }
    bool operator!=(const Vector2& o) const { return !(*this == o); }
};

struct Vector3 {
    float X = 0.0f, Y = 0.0f, Z = 0.0f;
    Vector3() {}
    Vector3(float x, float y, float z) : X(x), Y(y), Z(z) {}
    bool operator==(const Vector3& o) const { return X == o.X && Y == o.Y && Z == o.Z; }
    bool operator!=(const Vector3& o) const { return !(*this == o); }
    Vector3 operator+(const Vector3& o) const { return Vector3(X + o.X, Y + o.Y, Z + o.Z); }
    Vector3 operator-(const Vector3& o) const { return Vector3(X - o.X, Y - o.Y, Z - o.Z); }
    Vector3 operator*(float s) const { return Vector3(X * s, Y * s, Z * s); }
    Vector3 operator/(float s) const { return Vector3(X / s, Y / s, Z / s); }
};

struct Vertex {
    Vector3 Position;
    Vector3 Normal;
    Vector2 TextureCoordinate;
};

struct Material {
    std::string name;
    Vector3 Ka, Kd, Ks;
    float Ns = 0.0f, Ni = 0.0f, d = 0.0f;
    int illum = 0;
    std::string map_Ka, map_Kd, map_Ks, map_Ns, map_d, map_bump;
};

struct Mesh {
    std::string MeshName;
    std::vector<Vertex> Vertices;
    std::vector<unsigned int> Indices;   // always empty, as in the reference
    Material MeshMaterial;
};

class Loader {
public:
    bool LoadFile(std::string Path);

    std::vector<Mesh> LoadedMeshes;
    std::vector<Vertex> LoadedVertices;
    std::vector<unsigned int> LoadedIndices;
    std::vector<Material> LoadedMaterials;

private:
    bool LoadMaterials(const std::string& path);
};

// Triangulates one face (corner positions in order); appends local corner indices.  Exposed for tests.
void TriangulateFace(const std::vector<Vertex>& corners, std::vector<unsigned int>& out);

}  // namespace objl
