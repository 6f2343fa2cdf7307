#include "Model.h"

#include <cmath>
#include <cstring>
#include <iostream>

#include "OBJ_Loader.h"

static_assert(sizeof(Vertex) == sizeof(rt_vertex), "Vertex must keep the 76-byte layout the C ABI expects");

Model::Model(vec3 position, int size, const char* filename, Material mat, bool debug, ObjectType objectType) {
    this->position = position;
    this->size = size;
    this->mat = mat;
    this->debug = debug;
    this->objectType = objectType;
    objectName = Model_t;

    objl::Loader loader;
    if (!loader.LoadFile(filename)) {
        std::cerr << "Model: could not load " << filename << std::endl;
        return;
    }
    loaded = true;
    vertCount = (int)loader.LoadedVertices.size();
    indicesCount = (int)loader.LoadedIndices.size();
    materialCount = (int)loader.LoadedMaterials.size();
    meshCount = (int)loader.LoadedMeshes.size();
    std::cout << "Loaded " << filename << ": " << vertCount << " vertices, " << indicesCount << " indices, " << materialCount
              << " materials, " << meshCount << " meshes" << std::endl;

    // per-vertex inputs in LoadedVertices order; the mesh list repeats the same vertices mesh by mesh (Model.cpp:88-90)
    std::vector<float> pos((size_t)vertCount * 3), nrm((size_t)vertCount * 3), tex((size_t)vertCount * 2);
    std::vector<float> ka((size_t)vertCount * 3), kd((size_t)vertCount * 3), ks((size_t)vertCount * 3);
    size_t v = 0;
    for (const objl::Mesh& mesh : loader.LoadedMeshes) {
        const objl::Material& mm = mesh.MeshMaterial;
        for (const objl::Vertex& mv : mesh.Vertices) {
            if (v >= (size_t)vertCount) break;
            pos[3 * v] = mv.Position.X; pos[3 * v + 1] = mv.Position.Y; pos[3 * v + 2] = mv.Position.Z;
            nrm[3 * v] = mv.Normal.X; nrm[3 * v + 1] = mv.Normal.Y; nrm[3 * v + 2] = mv.Normal.Z;
            tex[2 * v] = mv.TextureCoordinate.X; tex[2 * v + 1] = mv.TextureCoordinate.Y;
            ka[3 * v] = mm.Ka.X; ka[3 * v + 1] = mm.Ka.Y; ka[3 * v + 2] = mm.Ka.Z;
            kd[3 * v] = mm.Kd.X; kd[3 * v + 1] = mm.Kd.Y; kd[3 * v + 2] = mm.Kd.Z;
            ks[3 * v] = mm.Ks.X; ks[3 * v + 1] = mm.Ks.Y; ks[3 * v + 2] = mm.Ks.Z;
            v++;
        }
    }
    // (The reference would also sample models/Atlas.png for materials with map_Kd, Model.cpp:105-112; that asset is
    //  not part of the checkout, and with the image missing the reference skips the lookup as well.)

    vertices.resize((size_t)vertCount);
    indices.resize((size_t)indicesCount);
    const int perBox = 3 * (int)std::ceil((double)(indicesCount / 3) / 8.0);
    std::vector<int32_t> chunk((size_t)perBox * 8 + 1);
    float bb[6], chunkBoxes[48];
    int32_t gotPerBox = 0;
    int rc = rt_bake_model(pos.data(), nrm.data(), tex.data(), ka.data(), kd.data(), ks.data(), vertCount,
                           loader.LoadedIndices.data(), indicesCount, this->position.nums, size,
                           reinterpret_cast<rt_vertex*>(vertices.data()), indices.data(), bb, chunkBoxes, chunk.data(), &gotPerBox);
    if (rc != RT_OK) {
        std::cerr << "Model: baking " << filename << " failed" << std::endl;
        vertices.clear(); indices.clear(); vertCount = indicesCount = 0; loaded = false;
        return;
    }
    boundingBox.min = vec3(bb[0], bb[1], bb[2]);
    boundingBox.max = vec3(bb[3], bb[4], bb[5]);
    for (int b = 0; b < 8; b++) {
        boundingBoxes[b].setupIndices(gotPerBox);
        std::memcpy(boundingBoxes[b].includedModelIndices.data(), chunk.data() + (size_t)b * gotPerBox, sizeof(int32_t) * (size_t)gotPerBox);
        boundingBoxes[b].min = vec3(chunkBoxes[6 * b], chunkBoxes[6 * b + 1], chunkBoxes[6 * b + 2]);
        boundingBoxes[b].max = vec3(chunkBoxes[6 * b + 3], chunkBoxes[6 * b + 4], chunkBoxes[6 * b + 5]);
    }

    // Model.cpp:119-124 (octree.init / octree.build, commented out in the reference): build the model's octree now, as the
    // flat arrays the renderer reads.  The pointer view is left out; the arrays passed to init are only used during build.
    octree = std::make_shared<Octree>();
    octree->pointerView = false;
    octree->init(toRt());
    octree->build(octree->root);
}

void Model::buildOctree(Octree& octree, std::vector<Vertex>& soup) const {
    soup.clear();
    for (int i = 0; i + 3 < indicesCount; i += 3) {          // the triangles Model.cpp:282 reaches
        for (int k = 0; k < 3; k++) soup.push_back(vertices[(size_t)indices[(size_t)i + k]]);
    }
    octree.init(soup.data(), (int)soup.size());
    octree.build(octree.root);
}

rt_model Model::toRt() const {
    rt_model m;
    std::memset(&m, 0, sizeof m);
    m.position = rt_vec3{position[0], position[1], position[2]};
    m.size = size;
    m.mat.ambientColour = rt_vec3{mat.ambientColour[0], mat.ambientColour[1], mat.ambientColour[2]};
    m.mat.diffuseColour = rt_vec3{mat.diffuseColour[0], mat.diffuseColour[1], mat.diffuseColour[2]};
    m.mat.ambient = mat.ambient; m.mat.diffuse = mat.diffuse; m.mat.specular = mat.specular; m.mat.shininess = mat.shininess;
    m.objectType = objectType;
    m.debug = debug ? 1 : 0;
    m.bbMin = rt_vec3{boundingBox.min[0], boundingBox.min[1], boundingBox.min[2]};
    m.bbMax = rt_vec3{boundingBox.max[0], boundingBox.max[1], boundingBox.max[2]};
    for (int b = 0; b < 8; b++) {
        const AABB& cb = boundingBoxes[b];
        m.boundingBoxes[b].min = rt_vec3{cb.min[0], cb.min[1], cb.min[2]};
        m.boundingBoxes[b].max = rt_vec3{cb.max[0], cb.max[1], cb.max[2]};
        m.boundingBoxes[b].indicesCount = cb.indicesCount;
        m.boundingBoxes[b].includedModelIndices = cb.includedModelIndices.empty() ? nullptr : cb.includedModelIndices.data();
    }
    m.vertices = reinterpret_cast<const rt_vertex*>(vertices.data());
    m.vertCount = vertCount;
    m.indices = indices.data();
    m.indicesCount = indicesCount;
    m.octree = (octree && octree->built() && octree->flat().nodeCount > 0) ? &octree->flat() : nullptr;
    return m;
}
