/*
 * src/sim5lib.h -- the path the reference's callers include (ref: examples/04-disk-image-eqplane/Makefile:2,
 * `SIM5LIB = ../../src`, `-I$(SIM5LIB)`).  The SIM5 API of this project lives in sim5_amd/host/sim5lib.h
 * (same names, structs and error conventions, over the MI355X library libsim5gpu.so); this file only puts it
 * where an unchanged reference example looks for it.
 */
#include "../sim5_amd/host/sim5lib.h"
