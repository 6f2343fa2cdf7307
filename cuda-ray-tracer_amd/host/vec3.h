// Host mirror of the reference's vec3 (vec3.h:6-118): three floats with the same accessors, operators and free
// functions, evaluated in the same order.  Plain C++, no device qualifiers: on this side of the C ABI the type only
// carries scene data; all ray arithmetic runs in the gfx950 kernels.
#pragma once
#include <cmath>

class vec3 {
public:
    float nums[3];

    vec3() : nums{0.0f, 0.0f, 0.0f} {}
    vec3(float a, float b, float c) : nums{a, b, c} {}

    float x() const { return nums[0]; }
    float y() const { return nums[1]; }
    float z() const { return nums[2]; }
    float r() const { return nums[0]; }
    float g() const { return nums[1]; }
    float b() const { return nums[2]; }

    float operator[](int i) const { return nums[i]; }
    float& operator[](int i) { return nums[i]; }
    vec3 operator-() const { return vec3(-nums[0], -nums[1], -nums[2]); }

    float length() const { return std::sqrt(nums[0] * nums[0] + nums[1] * nums[1] + nums[2] * nums[2]); }

    vec3& operator+=(const vec3& o) { for (int i = 0; i < 3; i++) nums[i] = nums[i] + o.nums[i]; return *this; }
    vec3& operator-=(const vec3& o) { for (int i = 0; i < 3; i++) nums[i] = nums[i] - o.nums[i]; return *this; }
    vec3& operator+=(int n) { for (int i = 0; i < 3; i++) nums[i] = nums[i] + n; return *this; }
    bool operator==(const vec3& o) const { return nums[0] == o.nums[0] && nums[1] == o.nums[1] && nums[2] == o.nums[2]; }
    bool operator!=(const vec3& o) const { return !(*this == o); }
};

inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline vec3 operator/(const vec3& a, const vec3& b) { return vec3(a[0] / b[0], a[1] / b[1], a[2] / b[2]); }
inline vec3 operator+(const vec3& a, float n) { return vec3(a[0] + n, a[1] + n, a[2] + n); }
inline vec3 operator-(const vec3& a, float n) { return vec3(a[0] - n, a[1] - n, a[2] - n); }
inline vec3 operator*(const vec3& a, float n) { return vec3(a[0] * n, a[1] * n, a[2] * n); }
inline vec3 operator*(float n, const vec3& a) { return vec3(a[0] * n, a[1] * n, a[2] * n); }
inline vec3 operator/(const vec3& a, float n) { return vec3(a[0] / n, a[1] / n, a[2] / n); }

inline float dot(const vec3& a, const vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline vec3 cross(const vec3& a, const vec3& b) {
    return vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline vec3 normalise(const vec3& v) { return v / v.length(); }
inline vec3 reflect(const vec3& a, const vec3& b) { return a - b * dot(a, b) * vec3(2.0f, 2.0f, 2.0f); }
