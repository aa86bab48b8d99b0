for w in C2 C3 C3pp C5pp; do for gflag in "" "--graph"; do for dt in bf16 fp32; do
echo -n "$w $dt $gflag: "; timeout 300 python bench.py --workload $w --dtype $dt $gflag --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['launch'])"
done; done; done
