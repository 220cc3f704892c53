python tools/landscape.py --genes 60,100,150 --ot 0.0,0.5,1.0 --reps 3 --ab 2>/dev/null
for g in 1 100; do GENES=$g python tools/ragged_rate.py 2>/dev/null | tail -1; done
