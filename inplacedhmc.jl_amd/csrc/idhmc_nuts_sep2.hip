// separable NUTS kernels, NCH = 5..8 (see idhmc_nuts_sep.inc)
#define IDHMC_NUTS_LO 5
#include "idhmc_nuts_sep.inc"
