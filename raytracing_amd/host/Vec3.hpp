/*
 * Vec3.hpp -- float[3] vector and 3x3 matrix with the reference's interface
 * (R/src/Vec3.h:12-119 Vec3, :121-288 Mat3).  Host-side scene setup only: the values built with
 * these types are what gets flattened and uploaded, so each operator keeps the reference's
 * operation order (sums left to right, true divides in normalize()).
 */
#pragma once
#include <cmath>
#include <iostream>

class Vec3 {
    float mVals[3];

public:
    Vec3() : mVals{0.f, 0.f, 0.f} {}
    Vec3(float x, float y, float z) : mVals{x, y, z} {}
    float& operator[](unsigned int c) { return mVals[c]; }
    float operator[](unsigned int c) const { return mVals[c]; }
    const float* data() const { return mVals; }

    float squareLength() const { return mVals[0] * mVals[0] + mVals[1] * mVals[1] + mVals[2] * mVals[2]; }
    float length() const { return std::sqrt(squareLength()); }
    float norm() const { return length(); }
    float squareNorm() const { return squareLength(); }
    void normalize() {
        float L = length();
        mVals[0] /= L; mVals[1] /= L; mVals[2] /= L;
    }
    static float dot(Vec3 const& a, Vec3 const& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static Vec3 cross(Vec3 const& a, Vec3 const& b) {
        return Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
    }
    static Vec3 compProduct(Vec3 const& a, Vec3 const& b) { return Vec3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
    /* The reference's clamp computes the clamped values and discards them (R/src/Vec3.h:87-93);
     * callers rely on getting the input back. */
    static Vec3 clamp(const Vec3& v, float, float) { return v; }

    void operator+=(Vec3 const& o) { mVals[0] += o[0]; mVals[1] += o[1]; mVals[2] += o[2]; }
    void operator-=(Vec3 const& o) { mVals[0] -= o[0]; mVals[1] -= o[1]; mVals[2] -= o[2]; }
    void operator*=(float s) { mVals[0] *= s; mVals[1] *= s; mVals[2] *= s; }
    void operator/=(float s) { mVals[0] /= s; mVals[1] /= s; mVals[2] /= s; }
};

inline Vec3 operator+(Vec3 const& a, Vec3 const& b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3 operator-(Vec3 const& a, Vec3 const& b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3 operator*(float a, Vec3 const& b) { return Vec3(a * b[0], a * b[1], a * b[2]); }
inline Vec3 operator*(Vec3 const& b, float a) { return Vec3(a * b[0], a * b[1], a * b[2]); }
inline Vec3 operator/(Vec3 const& a, float b) { return Vec3(a[0] / b, a[1] / b, a[2] / b); }
inline std::ostream& operator<<(std::ostream& s, Vec3 const& p) { return s << p[0] << " " << p[1] << " " << p[2]; }
inline std::istream& operator>>(std::istream& s, Vec3& p) { return s >> p[0] >> p[1] >> p[2]; }

class Mat3 {
    float vals[9];

public:
    Mat3() : vals{0, 0, 0, 0, 0, 0, 0, 0, 0} {}
    Mat3(float a, float b, float c, float d, float e, float f, float g, float h, float i) : vals{a, b, c, d, e, f, g, h, i} {}
    float operator()(unsigned int i, unsigned int j) const { return vals[3 * i + j]; }
    float& operator()(unsigned int i, unsigned int j) { return vals[3 * i + j]; }
    Vec3 operator*(const Vec3& p) const {
        const Mat3& m = *this;
        return Vec3(m(0, 0) * p[0] + m(0, 1) * p[1] + m(0, 2) * p[2], m(1, 0) * p[0] + m(1, 1) * p[1] + m(1, 2) * p[2],
                    m(2, 0) * p[0] + m(2, 1) * p[1] + m(2, 2) * p[2]);
    }
    Mat3 operator*(const Mat3& b) const {
        Mat3 r;
        for (unsigned i = 0; i < 3; ++i)
            for (unsigned j = 0; j < 3; ++j) r(i, j) = (*this)(i, 0) * b(0, j) + (*this)(i, 1) * b(1, j) + (*this)(i, 2) * b(2, j);
        return r;
    }
    Mat3 getTranspose() const { return Mat3(vals[0], vals[3], vals[6], vals[1], vals[4], vals[7], vals[2], vals[5], vals[8]); }
    float determinant() const {
        return vals[0] * (vals[4] * vals[8] - vals[7] * vals[5]) - vals[1] * (vals[3] * vals[8] - vals[6] * vals[5]) +
               vals[2] * (vals[3] * vals[7] - vals[6] * vals[4]);
    }
};
