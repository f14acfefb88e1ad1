for e in 0.25 0.03; do IDHMC_LIB=inplacedhmc.jl_amd/libidhmc_x1.so EPS=$e NT=5 python tools/bench_nuts.py 2>&1 | grep -E "steps/s"; done
for e in 0.25 0.03; do EPS=$e NT=5 python tools/bench_nuts.py 2>&1 | grep -E "steps/s"; done
