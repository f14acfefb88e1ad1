for v in ck3 ck2; do for e in 0.25 0.03; do echo -n "$v "; IDHMC_LIB=inplacedhmc.jl_amd/libidhmc_$v.so EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; done; done
for e in 0.25 0.03; do echo -n "base "; EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; done
for e in 0.25 0.03; do echo -n "base perchain "; METRIC=perchain EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; done
for e in 0.25 0.03; do echo -n "rich "; IDHMC_NUTS_WIDE=0 EPS=$e NT=10 python tools/bench_nuts.py 2>&1 | grep -E "steps/s" | cut -c1-110; done
rm -f gpurun_out/r03_nuts_bytes.jsonl
for e in 0.25 0.03; do IDHMC_LIB=inplacedhmc.jl_amd/libidhmc_bytes.so EPS=$e python tools/nuts_bytes.py 2>/dev/null >> gpurun_out/r03_nuts_bytes.jsonl; done
python -c "
import json
for l in open('gpurun_out/r03_nuts_bytes.jsonl'): d=json.loads(l); print(d['eps'], d['requested_bytes_per_leapfrog'], {k[:12]:round(v) for k,v in d['by_source_bytes_per_leapfrog'].items()})"
