"""ctypes mirror of include/rt_types.h (field order and widths must match the header)."""
import ctypes as C

c_i32 = C.c_int32
c_f32 = C.c_float


class Vec3(C.Structure):
    _fields_ = [("x", c_f32), ("y", c_f32), ("z", c_f32)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tuple(self):
        return (self.x, self.y, self.z)


def vec3(v):
    if isinstance(v, Vec3):
        return Vec3(v.x, v.y, v.z)
    return Vec3(v[0], v[1], v[2])


class Material(C.Structure):
    _fields_ = [("ambientColour", Vec3), ("diffuseColour", Vec3), ("ambient", c_f32), ("diffuse", c_f32),
                ("specular", c_f32), ("shininess", c_f32)]


def material(ka, kd, ambient, diffuse, specular, shininess):
    return Material(vec3(ka), vec3(kd), ambient, diffuse, specular, shininess)


class Light(C.Structure):
    _fields_ = [("position", Vec3), ("colour", Vec3)]


class Sphere(C.Structure):
    _fields_ = [("position", Vec3), ("radius", c_f32), ("mat", Material), ("objectType", c_i32), ("debug", c_i32)]


class Plane(C.Structure):
    _fields_ = [("position", Vec3), ("n", Vec3), ("mat", Material), ("objectType", c_i32), ("debug", c_i32)]


class AABB(C.Structure):
    _fields_ = [("min", Vec3), ("max", Vec3), ("mat", Material), ("objectType", c_i32), ("debug", c_i32)]


class Vertex(C.Structure):
    _fields_ = [("position", Vec3), ("textureCoords", Vec3), ("normal", Vec3), ("materialIndex", c_i32),
                ("ambient", Vec3), ("diffuse", Vec3), ("specular", Vec3)]


class ChunkBox(C.Structure):
    _fields_ = [("min", Vec3), ("max", Vec3), ("indicesCount", c_i32), ("includedModelIndices", C.POINTER(c_i32))]


class FlatOctree(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("nodeCount", C.c_int64), ("triangles", C.c_void_p), ("triangleRecords", C.c_int64),
                ("leafCount", c_i32), ("depth", c_i32), ("maxDepth", c_i32), ("leafTriangles", c_i32)]


class Model(C.Structure):
    _fields_ = [("position", Vec3), ("size", c_i32), ("mat", Material), ("objectType", c_i32), ("debug", c_i32),
                ("bbMin", Vec3), ("bbMax", Vec3), ("boundingBoxes", ChunkBox * 8),
                ("vertices", C.POINTER(Vertex)), ("vertCount", c_i32),
                ("indices", C.POINTER(c_i32)), ("indicesCount", c_i32), ("octree", C.POINTER(FlatOctree))]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("up", Vec3), ("right", Vec3), ("direction", Vec3), ("fovFactor", c_f32)]


class Config(C.Structure):
    _fields_ = [("fps", c_i32), ("shadowBias", c_f32), ("renderHardShadows", c_i32), ("renderSoftShadows", c_i32),
                ("softShadowRadius", c_i32), ("softShadowNum", c_i32), ("shadowIntensity", c_f32),
                ("areaLightSpecularEffect", c_i32), ("reflections", c_i32), ("maxDepth", c_i32),
                ("reflectPlanes", c_i32), ("sphereReflectionStrength", c_f32), ("planeReflectionStrength", c_f32),
                ("AABBReflectionStrength", c_f32), ("modelReflectionStrength", c_f32), ("renderAABBs", c_i32),
                ("renderModels", c_i32), ("ambientLighting", c_i32), ("diffuseLighting", c_i32),
                ("specularLighting", c_i32), ("antiAliasing", c_i32), ("boundingBox", c_i32),
                ("eightBoundingBoxes", c_i32), ("octree", c_i32), ("cullBackTriangles", c_i32),
                ("backgroundCol", Vec3), ("backgroundBrightness", c_i32), ("floorBrightness", c_i32)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("sphereCount", c_i32),
                ("planes", C.POINTER(Plane)), ("planeCount", c_i32),
                ("AABBs", C.POINTER(AABB)), ("AABBCount", c_i32),
                ("lights", C.POINTER(Light)), ("lightCount", c_i32),
                ("models", C.POINTER(Model)), ("modelCount", c_i32),
                ("octreeMaxDepth", c_i32), ("octreeLeafTriangles", c_i32), ("buildOnDevice", c_i32)]


class Dynamic(C.Structure):
    _fields_ = [("setLight", c_i32), ("lightPosition", Vec3),
                ("setSphere0", c_i32), ("sphere0Position", Vec3), ("sphere0Radius", c_f32),
                ("setPlane0", c_i32), ("plane0AmbientColour", Vec3)]


class FrameStats(C.Structure):
    _fields_ = [("gpu_ms", C.c_double), ("rays", C.c_uint64), ("node_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("attr_fetches", C.c_uint64), ("gate_tests", C.c_uint64), ("split_waves", C.c_uint64),
                ("node_visits", C.c_uint64), ("table_flops", C.c_uint64), ("useful_node_tests", C.c_uint64),
                ("useful_tri_tests", C.c_uint64), ("useful_node_visits", C.c_uint64)]


class SceneInfo(C.Structure):
    _fields_ = [("sphereCount", c_i32), ("planeCount", c_i32), ("AABBCount", c_i32), ("modelCount", c_i32),
                ("triangleCount", C.c_int64), ("nodeCount", C.c_int64), ("leafCount", C.c_int64),
                ("maxDepth", c_i32), ("deviceBytes", C.c_int64)]


RT_SPHERE_T, RT_PLANE_T, RT_BOX_T, RT_TRIANGLE_T, RT_MODEL_T, RT_AABB_T = range(6)
RT_DIFFUSE, RT_REFLECT = 0, 1
