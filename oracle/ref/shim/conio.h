// oracle/ref/shim: the reference includes <windows.h> from device-side headers; nothing of it is needed on this path
#ifndef BF_REF_SHIM_CONIO_H
#define BF_REF_SHIM_CONIO_H
#endif
