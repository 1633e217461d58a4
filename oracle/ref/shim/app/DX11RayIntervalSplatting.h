// oracle/ref/shim/app: the Direct3D 11 rasteriser pass of the ray caster (min / max ray interval per pixel).  CUDARayCastSDF.h holds one as
// a member; nothing compiled here calls it (the interval images of the pinned renderKernel are supplied by the glue, ref_raycast.cpp).
#pragma once
class DX11RayIntervalSplatting {};
