#!/bin/bash
# Re-tunes the launch knobs of the fused sweep kernel on the default workload (prints one line per setting).
b() { timeout -k 10 120 python bench.py --cpu-sweeps 0 "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; }
for s in 0 1 2 3 4 6; do echo "skew=$s $(ERM_SKEW=$s b)"; done
for w in 4 8 16; do echo "lanes=$w $(b --lanes-per-row $w)"; done
echo "grid=512x512 $(b --block-threads 512 --grid-blocks 512)"
echo "grid=512x1024 $(b --grid-blocks 512)"
echo "grid=128 $(b --grid-blocks 128)"
echo "default $(b)"
