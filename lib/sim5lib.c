/* lib/sim5lib.c -- see lib/sim5lib.h; thin include of the host side of the SIM5 API over libsim5gpu.so */
#include "../sim5_amd/host/sim5lib.c"
