for rep in 1 2; do
for v in ca cb; do echo -n "$v "; IDHMC_LIB=inplacedhmc.jl_amd/libidhmc_$v.so NUTS_ONLY=1 timeout -k 10 100 python tools/bench_dense.py 2>&1 | tail -1; done
echo -n "current "; NUTS_ONLY=1 python tools/bench_dense.py 2>&1 | tail -1
done
