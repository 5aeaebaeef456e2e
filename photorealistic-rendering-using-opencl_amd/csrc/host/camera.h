// camera.h -- orbit camera producing the 80-byte device Camera, API-compatible with the
// reference's InteractiveCamera (include/Camera/camera.h:17-56, src/Camera/camera.cpp).
#pragma once
#include "prt_types.h"

namespace prt {

using Camera = prt_camera;

class InteractiveCamera {
public:
    InteractiveCamera();                       // camera.cpp:4-12 defaults: yaw 0, pitch 0.3, radius 4, aperture 0.01, focal 4
    void changeYaw(float m);
    void changePitch(float m);
    void changeRadius(float m);
    void changeAltitude(float m);
    void changeFocalDistance(float m);
    void strafe(float m);
    void goForward(float m);
    void rotateRight(float m);
    void changeApertureDiameter(float m);
    void setResolution(float x, float y);
    void setFOVX(float fovx);
    void buildRenderCamera(Camera* renderCamera);

    float resolution[2] = {0.f, 0.f};
    float fov[2] = {0.f, 0.f};

private:
    float centerPosition[3] = {0.f, 0.f, 0.f};
    float viewDirection[3] = {0.f, 0.f, 0.f};
    float yaw, pitch, radius, apertureRadius, focalDistance;
};

}  // namespace prt
