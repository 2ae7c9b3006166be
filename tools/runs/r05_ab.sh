mkdir -p gpurun_out/r05t
B="python bench.py --no-cpu-baseline --no-reference-eager --no-alt-precision --no-roofline"
run() { local name=$1; shift; env "$@" timeout 200 $B > gpurun_out/r05t/$name.json 2> gpurun_out/r05t/$name.err; python - gpurun_out/r05t/$name.json $name <<'PY'
import json,sys
L=[l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith('{')]
d=json.loads(L[-1]); print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), flush=True)
PY
}
run base A=1
run g_overlap0 HG_G_OVERLAP=0
run wgrad_stream0 HG_WGRAD_STREAM=0
run base2 A=1
run wino_wgrad0 HG_WINO_WGRAD=0
run wino0 HG_WINO=0
run fused_dnl0 HG_FUSED_DNL=0
run graph1 HG_GRAPH=1
