// Host mirror of Camera (Camera.h:6-45, Camera.cpp:5-84).  The basis is computed by the library's rt_camera_look,
// which restates Camera::Camera / updateDirection, so that every host language gets identical floats.
#pragma once
#include "rt_api.h"
#include "utils.h"

class Camera {
private:
    vec3 up, right, direction, position;
    float fov = 90.0f, fovFactor = 1.0f, aspectRatio = 1.0f;
    float moveSpeed = 3.0f, sensitivity = 0.1f;

public:
    enum CameraMove { FORWARD, BACKWARD, LEFT, RIGHT, UP, DOWN };

    float pitch = 0.0f;
    float yaw = -90.0f;

    Camera() {}
    Camera(vec3 position, float fov, float aspectRatio, float yaw = -90, float pitch = 0)
        : position(position), fov(fov), aspectRatio(aspectRatio), pitch(pitch), yaw(yaw) { updateDirection(); }

    void updateDirection() {
        rt_camera c;
        rt_camera_look(position.nums, fov, yaw, pitch, &c);
        direction = vec3(c.direction.x, c.direction.y, c.direction.z);
        right = vec3(c.right.x, c.right.y, c.right.z);
        up = vec3(c.up.x, c.up.y, c.up.z);
        fovFactor = c.fovFactor;
    }

    // Camera.cpp:40-52
    void mouseMovement(double xOff, double yOff) {
        yaw += xOff * sensitivity;
        pitch += yOff * sensitivity;
        if (pitch > 89.0f) pitch = 89.0f;
        if (pitch < -89.0f) pitch = -89.0f;
        updateDirection();
    }

    // Camera.cpp:65-84 (world "up" on screen is -Y)
    void move(CameraMove m) {
        switch (m) {
            case FORWARD: position += direction * moveSpeed; break;
            case BACKWARD: position -= direction * moveSpeed; break;
            case LEFT: position -= right * moveSpeed; break;
            case RIGHT: position += right * moveSpeed; break;
            case UP: position[1] -= moveSpeed; break;
            case DOWN: position[1] += moveSpeed; break;
        }
    }

    vec3 getPosition() const { return position; }
    float getMoveSpeed() const { return moveSpeed; }
    float getSensitivity() const { return sensitivity; }

    // What the kernels read of a camera (Camera.cpp:22-38 uses right, up, direction, fovFactor).
    rt_camera toRt() const {
        rt_camera c;
        c.position = rt_vec3{position[0], position[1], position[2]};
        c.up = rt_vec3{up[0], up[1], up[2]};
        c.right = rt_vec3{right[0], right[1], right[2]};
        c.direction = rt_vec3{direction[0], direction[1], direction[2]};
        c.fovFactor = fovFactor;
        return c;
    }
};
