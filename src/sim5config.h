/*
 * src/sim5config.h -- the reference generates this file from sim5config.h.default at `make lib`
 * (ref: Makefile:29); callers that include it directly find the same macros here.  Host build: no CUDA.
 */
#ifndef _SIM5CONFIG_H
#define _SIM5CONFIG_H
#define DEVICEFUNC
#define HOSTFUNC
#define INLINE
#endif
