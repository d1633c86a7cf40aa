/*
 * src/sim5lib.c -- the translation unit the reference's example Makefile compiles
 * (ref: examples/04-disk-image-eqplane/Makefile:17, `$(SIM5LIB)/sim5lib.c` with gcc -O3 -w -fgnu89-inline, linked
 * with -lm only).  Thin include of the host side of the SIM5 API over libsim5gpu.so: no ray arithmetic here.
 */
#include "../sim5_amd/host/sim5lib.c"
