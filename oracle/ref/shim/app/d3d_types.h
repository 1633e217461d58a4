// oracle/ref/shim/app: the three Direct3D names GlobalAppState.h declares members with (never used by the files compiled here)
#pragma once
typedef long HRESULT;
struct ID3D11Device; struct ID3D11Query;
