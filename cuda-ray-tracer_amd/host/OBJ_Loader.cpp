#include "OBJ_Loader.h"

#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace {

// One float from the text at q, as strtof reads it (leading blanks, optional '+'), and q moved past it; q unmoved if
// there is none.  std::from_chars is several times faster than strtof and rounds the same way (both are correctly
// rounded); forms it does not take (hexadecimal, "inf", "nan") go to strtof.
inline float readFloat(char*& q) {
    char* s = q;
    while (*s == ' ' || *s == '\t') s++;
    const char* digits = (*s == '+') ? s + 1 : s;
    const char* end = digits;
    while (*end && *end != ' ' && *end != '\t' && *end != '\r' && *end != '\n' && *end != '/') end++;
    float v = 0.0f;
    const std::from_chars_result r = std::from_chars(digits, end, v);
    if (r.ec == std::errc() && r.ptr == end && !(*s == '+' && *digits == '-')) { q = const_cast<char*>(end); return v; }
    return std::strtof(q, &q);
}

}  // namespace

namespace objl {

namespace {

const char* skipBlank(const char* p) { while (*p == ' ' || *p == '\t') p++; return p; }
const char* skipToken(const char* p) { while (*p && *p != ' ' && *p != '\t') p++; return p; }

// text after the keyword, trimmed at both ends
std::string restOfLine(const char* afterKeyword) {
    const char* b = skipBlank(afterKeyword);
    const char* e = b + std::strlen(b);
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) e--;
    return std::string(b, e);
}

bool keywordIs(const char* tok, const char* tokEnd, const char* kw) {
    size_t n = std::strlen(kw);
    return (size_t)(tokEnd - tok) == n && std::memcmp(tok, kw, n) == 0;
}

Vector3 crossV(const Vector3& a, const Vector3& b) { return Vector3(a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X); }
float dotV(const Vector3& a, const Vector3& b) { return (a.X * b.X) + (a.Y * b.Y) + (a.Z * b.Z); }
float magV(const Vector3& a) { return std::sqrt(std::pow(a.X, 2.0f) + std::pow(a.Y, 2.0f) + std::pow(a.Z, 2.0f)); }

bool sameSide(const Vector3& p1, const Vector3& p2, const Vector3& a, const Vector3& b) {
    return dotV(crossV(b - a, p1 - a), crossV(b - a, p2 - a)) >= 0;
}

// OBJ_Loader.cpp:152-174: inside the prism of the triangle and with zero projection on its normal
bool insideTriangle(const Vector3& p, const Vector3& a, const Vector3& b, const Vector3& c) {
    if (!(sameSide(p, a, b, c) && sameSide(p, b, a, c) && sameSide(p, c, a, b))) return false;
    Vector3 n = crossV(b - a, c - a);
    Vector3 bn = n / magV(n);
    Vector3 proj = bn * dotV(p, bn);
    return magV(proj) == 0;
}

// One "v/vt/vn" corner reference -> resolved index (or -1 when the slot is absent).
template <typename T>
const T* resolve(const std::vector<T>& pool, long idx) {
    long i = idx < 0 ? (long)pool.size() + idx : idx - 1;
    if (i < 0 || i >= (long)pool.size()) return nullptr;
    return &pool[(size_t)i];
}

}  // namespace

// OBJ_Loader.cpp:576-741.  Corners are matched back to indices by POSITION equality, exactly as the reference does,
// so duplicate positions inside one polygon yield the same (odd) index lists the reference would produce.
void TriangulateFace(const std::vector<Vertex>& in, std::vector<unsigned int>& out) {
    if (in.size() < 3) return;
    if (in.size() == 3) { out.push_back(0); out.push_back(1); out.push_back(2); return; }

    std::vector<Vertex> work = in;
    auto emit = [&](const Vector3& first, const Vector3& second, const Vector3& third, size_t limit) {
        for (size_t j = 0; j < limit; j++) {
            if (in[j].Position == first) out.push_back((unsigned)j);
            if (in[j].Position == second) out.push_back((unsigned)j);
            if (in[j].Position == third) out.push_back((unsigned)j);
        }
    };
    for (;;) {
        for (int i = 0; i < (int)work.size(); i++) {
            const Vertex prev = work[i == 0 ? work.size() - 1 : (size_t)i - 1];
            const Vertex cur = work[(size_t)i];
            const Vertex next = work[(size_t)i == work.size() - 1 ? 0 : (size_t)i + 1];
            if (work.size() == 3) {
                emit(cur.Position, prev.Position, next.Position, work.size());   // the reference scans only 3 slots here
                work.clear();
                break;
            }
            if (work.size() == 4) {
                emit(cur.Position, prev.Position, next.Position, in.size());
                Vector3 fourth;
                for (const Vertex& w : work) {
                    if (w.Position != cur.Position && w.Position != prev.Position && w.Position != next.Position) { fourth = w.Position; break; }
                }
                emit(prev.Position, next.Position, fourth, in.size());
                work.clear();
                break;
            }
            // the interior-angle test of the reference can never reject (angle <= 0 && angle >= 180), so it is omitted
            bool blocked = false;
            for (const Vertex& v : in) {
                if (insideTriangle(v.Position, prev.Position, cur.Position, next.Position) && v.Position != prev.Position &&
                    v.Position != cur.Position && v.Position != next.Position) { blocked = true; break; }
            }
            if (blocked) continue;
            emit(cur.Position, prev.Position, next.Position, in.size());
            for (size_t j = 0; j < work.size(); j++) {
                if (work[j].Position == cur.Position) { work.erase(work.begin() + (long)j); break; }
            }
            i = -1;
        }
        if (out.empty() || work.empty()) break;
    }
}

bool Loader::LoadFile(std::string Path) {
    if (Path.size() < 4 || Path.substr(Path.size() - 4) != ".obj") return false;
    std::ifstream file(Path, std::ios::binary);
    if (!file.is_open()) return false;
    std::string text;
    file.seekg(0, std::ios::end);
    const std::streamoff bytes = file.tellg();
    if (bytes < 0) return false;
    text.resize((size_t)bytes);
    file.seekg(0, std::ios::beg);
    if (bytes > 0 && !file.read(&text[0], bytes)) return false;

    LoadedMeshes.clear(); LoadedVertices.clear(); LoadedIndices.clear();

    std::vector<Vector3> positions, normals;
    std::vector<Vector2> texcoords;
    std::vector<Vertex> meshVertices;       // vertices of the mesh being assembled
    size_t meshIndexCount = 0;
    std::vector<std::string> meshMaterialNames;
    std::string meshName;
    bool listening = false;
    std::vector<Vertex> corners;
    std::vector<unsigned int> local;

    auto flushMesh = [&](const std::string& name) {
        Mesh m;
        m.MeshName = name;
        m.Vertices.swap(meshVertices);
        LoadedMeshes.push_back(std::move(m));
        meshVertices.clear();
        meshIndexCount = 0;
    };

    // lines are cut in place: the line break (and a carriage return before it) becomes the terminator the number parsers need
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        char* const line = &text[pos];
        size_t lineLength = eol - pos;
        if (eol < text.size()) text[eol] = '\0';       // at the end of the text the string's own terminator follows
        pos = eol + 1;
        if (lineLength != 0 && line[lineLength - 1] == '\r') line[--lineLength] = '\0';
        const char* p = skipBlank(line);
        const char* kwEnd = skipToken(p);

        const bool isGroup = keywordIs(p, kwEnd, "o") || keywordIs(p, kwEnd, "g");
        if (isGroup || (lineLength != 0 && line[0] == 'g')) {
            const std::string newName = isGroup ? restOfLine(kwEnd) : std::string("unnamed");
            if (!listening) {
                listening = true;
                meshName = newName;
            } else if (meshIndexCount != 0 && !meshVertices.empty()) {
                flushMesh(meshName);
                meshName = restOfLine(kwEnd);
            } else {
                meshName = newName;
            }
        }
        if (keywordIs(p, kwEnd, "v")) {
            char* q = const_cast<char*>(kwEnd);
            Vector3 v;
            v.X = readFloat(q); v.Y = readFloat(q); v.Z = readFloat(q);
            positions.push_back(v);
        } else if (keywordIs(p, kwEnd, "vt")) {
            char* q = const_cast<char*>(kwEnd);
            Vector2 v;
            v.X = readFloat(q); v.Y = readFloat(q);
            texcoords.push_back(v);
        } else if (keywordIs(p, kwEnd, "vn")) {
            char* q = const_cast<char*>(kwEnd);
            Vector3 v;
            v.X = readFloat(q); v.Y = readFloat(q); v.Z = readFloat(q);
            normals.push_back(v);
        } else if (keywordIs(p, kwEnd, "f")) {
            corners.clear();
            bool missingNormal = false;
            Vertex carry;   // a corner without a slot keeps the previous corner's value, as the reference's reused temporary does
            const char* q = skipBlank(kwEnd);
            while (*q) {
                char* e;
                long vi = std::strtol(q, &e, 10), ti = 0, ni = 0;
                bool hasT = false, hasN = false;
                if (e == q) break;
                if (*e == '/') {
                    e++;
                    if (*e != '/' && *e != ' ' && *e != '\t' && *e) { char* e2; ti = std::strtol(e, &e2, 10); hasT = (e2 != e); e = e2; }
                    if (*e == '/') { e++; char* e2; ni = std::strtol(e, &e2, 10); hasN = (e2 != e); e = e2; }
                }
                const Vector3* P = resolve(positions, vi);
                if (!P) return false;
                carry.Position = *P;
                if (hasT) { const Vector2* T = resolve(texcoords, ti); if (!T) return false; carry.TextureCoordinate = *T; }
                else carry.TextureCoordinate = Vector2(0, 0);
                if (hasN) { const Vector3* N = resolve(normals, ni); if (!N) return false; carry.Normal = *N; }
                else missingNormal = true;
                corners.push_back(carry);
                q = skipBlank(skipToken(e));
            }
            if (missingNormal && corners.size() >= 3) {
                Vector3 n = crossV(corners[0].Position - corners[1].Position, corners[2].Position - corners[1].Position);
                for (Vertex& c : corners) c.Normal = n;
            }
            for (const Vertex& c : corners) { meshVertices.push_back(c); LoadedVertices.push_back(c); }
            local.clear();
            TriangulateFace(corners, local);
            const unsigned base = (unsigned)(LoadedVertices.size() - corners.size());
            for (unsigned li : local) LoadedIndices.push_back(base + li);
            meshIndexCount += local.size();
        } else if (keywordIs(p, kwEnd, "usemtl")) {
            meshMaterialNames.push_back(restOfLine(kwEnd));
            if (meshIndexCount != 0 && !meshVertices.empty()) flushMesh(meshName + "_2");   // OBJ_Loader.cpp:369-377
        } else if (keywordIs(p, kwEnd, "mtllib")) {
            size_t slash = Path.find_last_of('/');
            std::string dir = slash == std::string::npos ? std::string() : Path.substr(0, slash + 1);
            LoadMaterials(dir + restOfLine(kwEnd));
        }
    }
    if (meshIndexCount != 0 && !meshVertices.empty()) flushMesh(meshName);

    for (size_t i = 0; i < meshMaterialNames.size() && i < LoadedMeshes.size(); i++) {
        for (const Material& m : LoadedMaterials) {
            if (m.name == meshMaterialNames[i]) { LoadedMeshes[i].MeshMaterial = m; break; }
        }
    }
    return !(LoadedMeshes.empty() && LoadedVertices.empty() && LoadedIndices.empty());
}

// OBJ_Loader.cpp:743-902
bool Loader::LoadMaterials(const std::string& path) {
    if (path.size() < 4 || path.substr(path.size() - 4) != ".mtl") return false;
    std::ifstream file(path);
    if (!file.is_open()) return false;
    Material cur;
    bool listening = false;
    std::string line;
    while (std::getline(file, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char* p = skipBlank(line.c_str());
        const char* kwEnd = skipToken(p);
        auto three = [&](Vector3& dst) {
            char* q = const_cast<char*>(kwEnd);
            char* e;
            float v[3]; int n = 0;
            for (; n < 3; n++) { v[n] = std::strtof(q, &e); if (e == q) break; q = e; }
            if (n == 3 && *skipBlank(q) == 0) dst = Vector3(v[0], v[1], v[2]);
        };
        if (keywordIs(p, kwEnd, "newmtl")) {
            if (listening) { LoadedMaterials.push_back(cur); cur = Material(); }
            listening = true;
            cur.name = line.size() > 7 ? restOfLine(kwEnd) : std::string("none");
        } else if (keywordIs(p, kwEnd, "Ka")) three(cur.Ka);
        else if (keywordIs(p, kwEnd, "Kd")) three(cur.Kd);
        else if (keywordIs(p, kwEnd, "Ks")) three(cur.Ks);
        else if (keywordIs(p, kwEnd, "Ns")) cur.Ns = std::strtof(kwEnd, nullptr);
        else if (keywordIs(p, kwEnd, "Ni")) cur.Ni = std::strtof(kwEnd, nullptr);
        else if (keywordIs(p, kwEnd, "d")) cur.d = std::strtof(kwEnd, nullptr);
        else if (keywordIs(p, kwEnd, "illum")) cur.illum = (int)std::strtol(kwEnd, nullptr, 10);
        else if (keywordIs(p, kwEnd, "map_Ka")) cur.map_Ka = restOfLine(kwEnd);
        else if (keywordIs(p, kwEnd, "map_Kd")) cur.map_Kd = restOfLine(kwEnd);
        else if (keywordIs(p, kwEnd, "map_Ks")) cur.map_Ks = restOfLine(kwEnd);
        else if (keywordIs(p, kwEnd, "map_Ns")) cur.map_Ns = restOfLine(kwEnd);
        else if (keywordIs(p, kwEnd, "map_d")) cur.map_d = restOfLine(kwEnd);
        else if (keywordIs(p, kwEnd, "map_Bump") || keywordIs(p, kwEnd, "map_bump") || keywordIs(p, kwEnd, "bump")) cur.map_bump = restOfLine(kwEnd);
    }
    LoadedMaterials.push_back(cur);   // the last (or only, possibly empty) material is always pushed
    return true;
}

}  // namespace objl
