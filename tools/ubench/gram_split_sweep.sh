#!/bin/bash
# development: sweep the K split of the Gram kernel (TLSQ_GRAM_SPLIT=o,d) at one size;  gram_split_sweep.sh M N "o,d o,d ..."
M=$1; N=$2
echo "model: $(TLSQ_DEBUG=2 python tools/kbench.py gram --M $M --N $N --reps 5 2>&1 | grep -m1 "nsplit") -> $(python tools/kbench.py gram --M $M --N $N --reps 5 | tail -1)"
for s in $3; do echo "split $s: $(TLSQ_GRAM_SPLIT=$s python tools/kbench.py gram --M $M --N $N --reps 5 | tail -1)"; done
