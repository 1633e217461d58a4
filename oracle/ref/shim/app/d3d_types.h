// oracle/ref/shim/app: the Direct3D / DXUT names the application headers declare members and error paths with (never reached by the files compiled here)
#pragma once
typedef long HRESULT;
struct ID3D11Device; struct ID3D11Query; struct ID3D11DeviceContext;
#ifndef S_OK
#define S_OK 0
#endif
#ifndef V_RETURN
#define V_RETURN(x) { hr = (x); if (hr < 0) return hr; }
#endif
inline ID3D11DeviceContext* DXUTGetD3D11DeviceContext() { return nullptr; }
inline ID3D11Device* DXUTGetD3D11Device() { return nullptr; }
