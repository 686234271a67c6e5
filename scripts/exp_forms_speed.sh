for B in 64 256; do
  L=3; [ $B = 256 ] && L=2
  for cfg in "4 128" "244 128" "4 256"; do
    set -- $cfg
    POSERISK_WINOGRAD=$1 POSERISK_WINOGRAD_MIN_C=$2 timeout -k 10 300 python bench.py --batch $B --lanes $L --cpu-frames 0 --steps 30 > gpurun_out/r03_form.json 2>/dev/null
    python -c "
import json;d=json.load(open('gpurun_out/r03_form.json'));print('B=$B form=$1 min_c=$2', d['value'], d['value_spread']['median'], d['frames_per_s_one_batch_in_flight'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])"
  done
done
