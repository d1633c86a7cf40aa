/*
 * lib/sim5lib.h -- where the reference's `make export` puts the single-file library
 * (ref: Makefile:57-67; used by examples/01-kerr-spacetime/Makefile:2, `SIM5LIB = ../../lib`).
 */
#include "../sim5_amd/host/sim5lib.h"
