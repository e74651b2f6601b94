// Drop-in for s2899221/Code/persp_camera.h: the declarations live in rt_host.hpp.
#ifndef RT_HOST_PERSP_CAMERA_H
#define RT_HOST_PERSP_CAMERA_H
#include "rt_host.hpp"
#endif
