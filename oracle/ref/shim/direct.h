// oracle/ref/shim: <direct.h> (_mkdir) is not needed
