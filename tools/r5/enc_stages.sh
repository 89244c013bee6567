#!/bin/bash
# Runs ON THE GPU BOX: value, encode rate and stage times of the headline (twice), the 4K key-frame share and the I+P workload
one() { python3 bench.py "$@" --no-cpu --no-host-boundary 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['config']['enc_MPix_s_rank0'],d['config']['stage_ms_per_step'])"; }
one --no-others --steps 5 --warmup 1; one --no-others --steps 5 --warmup 1
one --width 3840 --height 2160 --frames 150 --steps 3 --warmup 1
one --workload ip --gop 50 --steps 2 --warmup 1
