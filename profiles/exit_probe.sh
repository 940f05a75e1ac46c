cd $GRAFT_REPO_ROOT
python3 - <<'P'
import json
d=json.load(open("mm2-gb_amd/mi355x_config.json")); d["num_streams"]=16
json.dump(d,open("/tmp/cfg16.json","w"))
d["num_streams"]=1
json.dump(d,open("/tmp/cfg1.json","w"))
P
H=oracle/_ref/minimap2_gpuhost_rmq
T=tests/golden/data
for cfg in 1 16; do
 for mode in park now; do
  s=$(date +%s.%N)
  MM2GB_FREE=$mode MM2GB_DEBUG_PHASES=1 $H -t $cfg --max-chain-skip=2147483647 --gpu-chain --gpu-cfg /tmp/cfg$cfg.json $T/MT-human.fa $T/MT-orang.fa > /dev/null 2> /tmp/err.txt
  e=$(date +%s.%N)
  echo "streams=$cfg free=$mode whole=$(python3 -c "print(round($e - $s, 3))") :: $(grep -E 'init_stream_gpu|free_stream_gpu' /tmp/err.txt | sed 's/.*epoch [0-9.]*, //' | tr '\n' ';') start_epoch=$s end_epoch=$e $(grep -oE 'entered at epoch [0-9.]+' /tmp/err.txt | tr '\n' ' ')"
 done
done
