// Drop-in for s2899221/Code/raytracer.h: the declarations live in rt_host.hpp.
#ifndef RT_HOST_RAYTRACER_H
#define RT_HOST_RAYTRACER_H
#include "rt_host.hpp"
#endif
