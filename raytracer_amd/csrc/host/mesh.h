// Drop-in for s2899221/Code/mesh.h: the declarations live in rt_host.hpp.
#ifndef RT_HOST_MESH_H
#define RT_HOST_MESH_H
#include "rt_host.hpp"
#endif
