#!/bin/bash
# in-box A/B of the V-JEPA step (bench.py --vjepa): tools/vjepa_ab.sh <variant> ...  (smbv_debug_set_variant bits; "dev" = index arrays derived on the device
# by the forward instead of shipped by the collate function)
for P in 1 2; do for V in "$@"; do
  if [ "$V" = dev ]; then E=1; B=0; else E=0; B=$V; fi
  SMBV_VJEPA_DEVICE_PLUMBING=$E python bench.py --vjepa --steps 6 --warmup 2 --debug-variant $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $P vjepa variant $V', round(d['ms_per_step'],2))"
done; done
