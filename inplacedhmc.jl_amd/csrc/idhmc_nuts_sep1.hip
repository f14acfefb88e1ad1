// separable NUTS kernels, NCH = 1..4 (see idhmc_nuts_sep.inc)
#define IDHMC_NUTS_LO 1
#include "idhmc_nuts_sep.inc"
