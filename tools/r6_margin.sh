#!/bin/bash
# exploration for the HBM-margin test: map a golden set with a foreign process holding all but KEEP MB of the device
G=tests/golden; SET=${SET:-g6_repeats}
mkdir -p /tmp/mg && for f in $G/$SET/*.gz; do gunzip -c $f > /tmp/mg/$(basename ${f%.gz}); done; ls /tmp/mg | head
for keep in ${KEEPS:-100000 4000 3000 2600 2300 2100}; do
  mkfifo /tmp/mg/fifo_$keep 2>/dev/null
  ( sleep 1000 > /tmp/mg/fifo_$keep & echo $! > /tmp/mg/sl_$keep ) 
  python3 tests/helpers/hbm_fill.py $keep < /tmp/mg/fifo_$keep > /tmp/mg/fill_$keep.txt 2>&1 &
  FP=$!
  for i in $(seq 1 100); do grep -q ready /tmp/mg/fill_$keep.txt 2>/dev/null && break; sleep 0.2; done
  echo "== keep $keep MB: $(cat /tmp/mg/fill_$keep.txt)"
  timeout 120 env AL_TIMING=1 AL_PG_PLAIN=1 ${ENVX:-} airlift_amd/bin/airlift-align -ax sr -t 8 /tmp/mg/rep.fa /tmp/mg/g6_1.fq /tmp/mg/g6_2.fq > /tmp/mg/out_$keep.sam 2> /tmp/mg/err_$keep.txt; echo "   exit $? $(grep -c . /tmp/mg/out_$keep.sam) lines; $(grep -ci halv /tmp/mg/err_$keep.txt) halved line(s)"
  grep -iE "halv|failed|NOMEM|out of|margin" /tmp/mg/err_$keep.txt | head -5 | cut -c1-200
  cmp -s /tmp/mg/out_$keep.sam /tmp/mg/out_100000.sam && echo "   identical to the unconstrained run"
  kill $(cat /tmp/mg/sl_$keep) 2>/dev/null; wait $FP 2>/dev/null
done
