// oracle/ref/shim: CUDAImageUtil.cu spells the include "mlibCuda.h" (case-insensitive file system); forward to the reference header
#include "mLibCuda.h"
