// oracle/ref/shim: the application's precompiled header (mLib, DirectX, ...) is not needed by the files compiled here
