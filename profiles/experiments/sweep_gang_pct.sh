# the share of the launch's workgroups a gang chunk gets (MM2GB_GANG_PCT) per batch size (MM2GB_GANG_CAP: a cap on what all gangs hold, built for
# this sweep and not kept -- the library ignores it; results: profiles/r04_gang_knobs.txt):
# bench.py kernel-only step (window + plan + score), mixed 100-300 kb reads; then the two long-read bins at 100 M anchors
run() { env $2 timeout 200 python bench.py $1 --steps 6 --warmup 2 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e12,3))"; }
CFGS=("MM2GB_GANG_PCT=150 MM2GB_GANG_CAP=100" "MM2GB_GANG_PCT=150 MM2GB_GANG_CAP=85" "MM2GB_GANG_PCT=250 MM2GB_GANG_CAP=85" "MM2GB_GANG_PCT=400 MM2GB_GANG_CAP=85" "MM2GB_GANG_PCT=250 MM2GB_GANG_CAP=70" "MM2GB_GANG_PCT=400 MM2GB_GANG_CAP=70" "MM2GB_GANG_PCT=400 MM2GB_GANG_CAP=55")
for n in 20000000 50000000 100000000 150000000; do for cfg in "${CFGS[@]}"; do echo "$n | $cfg | $(run "--anchors $n" "$cfg")"; done; done
for lens in "200000 300000" "100000 200000"; do set -- $lens; for cfg in "${CFGS[@]}"; do echo "100M $1-$2 | $cfg | $(run "--anchors 100000000 --len-lo $1 --len-hi $2" "$cfg")"; done; done
