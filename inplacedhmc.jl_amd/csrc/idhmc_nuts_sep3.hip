// separable NUTS kernels, NCH = 9..12 (see idhmc_nuts_sep.inc)
#define IDHMC_NUTS_LO 9
#include "idhmc_nuts_sep.inc"
