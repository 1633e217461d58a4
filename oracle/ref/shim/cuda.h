// oracle/ref/shim: forwards to the host stand-in (test infrastructure only)
#include "cuda_runtime.h"
