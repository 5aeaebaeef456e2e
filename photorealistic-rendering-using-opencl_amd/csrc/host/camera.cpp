// camera.cpp -- see camera.h.  Float/double promotion follows the reference expression by
// expression (src/Camera/camera.cpp:68-104) so the uploaded 80 bytes are identical; checked
// against the reference's own camera code in tests/test_host_scene.py.
#include "camera.h"

#include <cmath>
#include <cstring>

namespace prt {

static const double kPi = 3.14159265358979323846;      // M_PI
static const double kPiOverTwo = 1.5707963267948966192313216916397514420985;

InteractiveCamera::InteractiveCamera() {
    yaw = 0;
    pitch = 0.3;
    radius = 4;
    apertureRadius = 0.01;
    focalDistance = 4.0f;
}

static float wrap(float x, float y) { return x - y * floorf(x / y); }
static float clamp2(float n, float lo, float hi) { n = fminf(n, hi); n = fmaxf(n, lo); return n; }

void InteractiveCamera::changeYaw(float m) { yaw += m; yaw = wrap(yaw, (float)(2 * kPi)); }
void InteractiveCamera::changePitch(float m) {
    pitch += m;
    float padding = 0.05;
    pitch = clamp2(pitch, (float)(-kPiOverTwo + padding), (float)(kPiOverTwo - padding));
}
void InteractiveCamera::changeRadius(float m) { radius += radius * m; radius = clamp2(radius, 0.2f, 100.0f); }
void InteractiveCamera::changeAltitude(float m) { centerPosition[1] += m; }
void InteractiveCamera::goForward(float m) { for (int i = 0; i < 3; ++i) centerPosition[i] += viewDirection[i] * m; }
void InteractiveCamera::strafe(float m) {
    // cross(viewDirection, (0,1,0)) normalised
    float ax = viewDirection[1] * 0.0f - viewDirection[2] * 1.0f;
    float ay = viewDirection[2] * 0.0f - viewDirection[0] * 0.0f;
    float az = viewDirection[0] * 1.0f - viewDirection[1] * 0.0f;
    float n = sqrtf(ax * ax + ay * ay + az * az);
    centerPosition[0] += (ax / n) * m; centerPosition[1] += (ay / n) * m; centerPosition[2] += (az / n) * m;
}
void InteractiveCamera::rotateRight(float m) {
    float yaw2 = yaw + m;
    float x = sinf(yaw2) * cosf(pitch), y = sinf(pitch), z = cosf(yaw2) * cosf(pitch);
    viewDirection[0] = (float)(x * (-1.0)); viewDirection[1] = (float)(y * (-1.0)); viewDirection[2] = (float)(z * (-1.0));
}
void InteractiveCamera::changeApertureDiameter(float m) {
    apertureRadius += (apertureRadius + 0.01) * m;
    apertureRadius = clamp2(apertureRadius, 0.0f, 25.0f);
}
void InteractiveCamera::changeFocalDistance(float m) { focalDistance += m; focalDistance = clamp2(focalDistance, 0.2f, 100.0f); }
void InteractiveCamera::setResolution(float x, float y) { resolution[0] = x; resolution[1] = y; }

void InteractiveCamera::setFOVX(float fovx) {
    // camera.cpp:72-86: degreesToRadians/radiansToDegrees take and return float, compute in double
    fov[0] = fovx;
    float rad = (float)(fovx / 180.0 * kPi);
    float inner = (float)(atan(tan(rad * 0.5) * (resolution[1] / resolution[0])) * 2.0);
    fov[1] = (float)(inner * 180.0 / kPi);
}

void InteractiveCamera::buildRenderCamera(Camera* c) {
    // camera.cpp:88-104.  <cmath> overloads: sin(float) etc. are float functions.
    float xDirection = sinf(yaw) * cosf(pitch);
    float yDirection = sinf(pitch);
    float zDirection = cosf(yaw) * cosf(pitch);
    float dirToCam[3] = {xDirection, yDirection, zDirection};
    std::memset(c, 0, sizeof(*c));
    for (int i = 0; i < 3; ++i) {
        viewDirection[i] = (float)(dirToCam[i] * (-1.0));   // vec3 * double -> operator*(float)
        c->position[i] = centerPosition[i] + dirToCam[i] * radius;
        c->view[i] = viewDirection[i];
    }
    c->view[3] = -0.0f;                                     // vec4(x,y,z,0) * -1: the unused w lane is -0
    c->up[0] = 0.f; c->up[1] = 1.f; c->up[2] = 0.f;
    c->resolution[0] = resolution[0]; c->resolution[1] = resolution[1];
    c->fov[0] = fov[0]; c->fov[1] = fov[1];
    c->apertureRadius = apertureRadius;
    c->focalDistance = focalDistance;
}

}  // namespace prt
