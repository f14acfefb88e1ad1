// separable NUTS kernels, NCH = 13..16 (see idhmc_nuts_sep.inc)
#define IDHMC_NUTS_LO 13
#include "idhmc_nuts_sep.inc"
