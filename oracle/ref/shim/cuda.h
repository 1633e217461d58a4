// oracle/ref/shim: forwards to the host stand-in (test infrastructure only)
#include "cuda_runtime.h"
#ifndef CUDA_VERSION
#define CUDA_VERSION 7000      /* the toolkit generation the reference was built with */
#endif
