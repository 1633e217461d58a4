// oracle/ref/shim/app: see SiftGPU/MatrixConversion.h next to this file
#include "SiftGPU/MatrixConversion.h"
