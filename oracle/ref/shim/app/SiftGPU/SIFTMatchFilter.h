// oracle/ref/shim/app: CPU reference filters of the SiftGPU fork (mLib images); Bundler.cpp includes the header and uses nothing of it
