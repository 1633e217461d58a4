// oracle/ref/shim/app: debug image output (mLib images, FreeImage); SBA.cpp includes it and uses nothing of it
