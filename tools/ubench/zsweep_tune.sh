#!/bin/bash
# Tile shape of the E-free sweep at C2 size: rows per thread x column tile (TLSQ_RUS_ROWS / TLSQ_RUS_CT), phases of one solve.
for rows in 1 2; do for ct in 8 16 32 64; do
  echo "rows=$rows ct=$ct: $(TLSQ_RUS_ROWS=$rows TLSQ_RUS_CT=$ct python tools/c2_debug.py ${1:-20000} ${2:-512} ${3:-16} 2>/dev/null | tail -1 | sed 's/.*loop=/loop=/')"
done; done
