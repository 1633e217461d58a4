// oracle/ref/shim/app: Direct3D interop is not part of the path (CUDASolverBundling.h includes it and uses nothing of it)
