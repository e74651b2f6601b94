// Drop-in for s2899221/Code/utils.h: the declarations live in rt_host.hpp.
#ifndef RT_HOST_UTILS_H
#define RT_HOST_UTILS_H
#include "rt_host.hpp"
#endif
