/*
 * Camera.hpp -- the reference's Camera interface (R/src/Camera.h:15-57, R/src/Camera.cpp:24-145)
 * without an OpenGL context.  The reference stores its camera in the GL matrix stack and the
 * renderer reads the PROJECTION / MODELVIEW matrices back (R/src/matrixUtilities.h:188-217); here
 * resize() and apply() write the same two float matrices into the object:
 *   projection = gluPerspective(fovy, aspect, near, far): double arithmetic (SGI libGLU), stored
 *                as float by the GL
 *   modelview  = identity, glTranslatef(x, y, z), glTranslatef(0, 0, -zoom), glMultMatrixf(rot),
 *                each step in float as Mesa's matrix code does it
 * glState() hands them to the C-ABI.
 */
#pragma once
#include "../../include/rtamd.h"
#include "Trackball.hpp"
#include "Vec3.hpp"

class Camera {
public:
    Camera() {
        fovAngle = 45.0; aspectRatio = 1.0; nearPlane = 4.1; farPlane = 10000.0;
        spinning = moving = beginu = beginv = 0;
        H = W = 1;
        trackball(curquat, 0.0, 0.0, 0.0, 0.0);
        x = y = z = 0.0;
        _zoom = 3.0;
        for (int i = 0; i < 16; ++i) projection[i] = modelview[i] = (i % 5 == 0) ? 1.f : 0.f;
    }
    virtual ~Camera() {}

    float getFovAngle() const { return fovAngle; }
    void setFovAngle(float a) { fovAngle = a; }
    float getAspectRatio() const { return aspectRatio; }
    float getNearPlane() const { return nearPlane; }
    void setNearPlane(float n) { nearPlane = n; }
    float getFarPlane() const { return farPlane; }
    void setFarPlane(float f) { farPlane = f; }
    unsigned int getScreenWidth() const { return W; }
    unsigned int getScreenHeight() const { return H; }

    void resize(int _W, int _H) { /* Camera.cpp:41-50 */
        H = _H; W = _W;
        aspectRatio = static_cast<float>(W) / static_cast<float>(H);
        const double pi = 3.14159265358979323846;
        double fovy = fovAngle, aspect = aspectRatio, zNear = nearPlane, zFar = farPlane;
        double radians = fovy / 2 * pi / 180, deltaZ = zFar - zNear, sine = std::sin(radians);
        if (deltaZ == 0 || sine == 0 || aspect == 0) return;
        double cotangent = std::cos(radians) / sine;
        for (int i = 0; i < 16; ++i) projection[i] = 0.f;
        projection[0] = (float)(cotangent / aspect);
        projection[5] = (float)cotangent;
        projection[10] = (float)(-(zFar + zNear) / deltaZ);
        projection[11] = -1.f;
        projection[14] = (float)(-2 * zNear * zFar / deltaZ);
    }
    void initPos() {}
    void move(float dx, float dy, float dz) { x += dx; y += dy; z += dz; }
    void beginRotate(int u, int v) { beginu = u; beginv = v; moving = 1; spinning = 0; }
    void rotate(int u, int v) { /* Camera.cpp:100-112 */
        if (!moving) return;
        trackball(lastquat, (2.0 * beginu - W) / W, (H - 2.0 * beginv) / H, (2.0 * u - W) / W, (H - 2.0 * v) / H);
        beginu = u; beginv = v; spinning = 1;
        add_quats(lastquat, curquat, curquat);
    }
    void endRotate() { moving = false; }
    void zoom(float dz) { _zoom += dz; }
    /* not in the reference: set the trackball quaternion directly (headless stand-in for mouse drags) */
    void setRotation(const float q[4]) { for (int i = 0; i < 4; ++i) curquat[i] = q[i]; }

    void apply() { /* Camera.cpp:125-132 */
        float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        translate(m, x, y, z);
        float rot[4][4];
        build_rotmatrix(rot, curquat);
        translate(m, 0.0f, 0.0f, -_zoom);
        const float* r = &rot[0][0];
        for (int c = 0; c < 4; ++c)
            for (int rr = 0; rr < 4; ++rr)
                modelview[4 * c + rr] = m[rr] * r[4 * c] + m[4 + rr] * r[4 * c + 1] + m[8 + rr] * r[4 * c + 2] + m[12 + rr] * r[4 * c + 3];
    }
    void getPos(float& X, float& Y, float& Z) { /* Camera.cpp:135-145 */
        float m[4][4];
        build_rotmatrix(m, curquat);
        float px = -x, py = -y, pz = -z + _zoom;
        X = m[0][0] * px + m[0][1] * py + m[0][2] * pz;
        Y = m[1][0] * px + m[1][1] * py + m[1][2] * pz;
        Z = m[2][0] * px + m[2][1] * py + m[2][2] * pz;
    }
    void getPos(Vec3& p) { getPos(p[0], p[1], p[2]); }

    /* the matrices the renderer reads back, plus GL_DEPTH_RANGE near (0 by default) */
    rt_camera glState() const {
        rt_camera c;
        for (int i = 0; i < 16; ++i) { c.projection[i] = projection[i]; c.modelview[i] = modelview[i]; }
        c.depth_near = 0.f;
        return c;
    }

private:
    static void translate(float m[16], float tx, float ty, float tz) {
        for (int r = 0; r < 4; ++r) m[12 + r] = m[r] * tx + m[4 + r] * ty + m[8 + r] * tz + m[12 + r];
    }
    float fovAngle, aspectRatio, nearPlane, farPlane;
    int spinning, moving, beginu, beginv, H, W;
    float curquat[4], lastquat[4];
    float x, y, z, _zoom;
    float projection[16], modelview[16];
};
