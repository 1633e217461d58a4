// oracle/ref/shim: math_constants.h names used by the reference (test infrastructure only)
#ifndef BF_REF_SHIM_MATH_CONSTANTS_H
#define BF_REF_SHIM_MATH_CONSTANTS_H
#include "cuda_runtime.h"
#define CUDART_INF_F __int_as_float(0x7f800000)
#define CUDART_NAN_F __int_as_float(0x7fffffff)
#define CUDART_PI_F 3.141592654f
#define CUDART_PIO2_F 1.570796327f
#define CUDART_SQRT_TWO_F 1.414213562f
#define CUDART_MAX_NORMAL_F __int_as_float(0x7f7fffff)
#define CUDART_MIN_DENORM_F __int_as_float(0x00000001)
#endif
