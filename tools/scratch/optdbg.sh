cd /root/repo/tests/golden/g6_repeats && mkdir -p /tmp/g6 && for f in *.gz; do zcat $f > /tmp/g6/${f%.gz}; done; cd /tmp/g6
REF=$(python3 -c "import json;print(json.load(open('/root/repo/tests/golden/g6_repeats/meta.json'))['ref'])"); R=$(python3 -c "import json;print(' '.join(json.load(open('/root/repo/tests/golden/g6_repeats/meta.json'))['reads']))")
/root/repo/oracle/_ref/mm2ref -g 300 -F 1200 -r 50 $REF $R > exp.sam 2>/dev/null
for d in 0 2147483648 536870912; do AL_DBG=$d /root/repo/airlift_amd/bin/airlift-align -ax sr -g 300 -F 1200 -r 50 $REF $R > got.sam 2>/dev/null; echo "AL_DBG=$d: $(diff got.sam exp.sam | grep -c '^<') differing"; done
for o in "-r 50" "-g 300" "-F 1200"; do /root/repo/oracle/_ref/mm2ref $o $REF $R > exp.sam 2>/dev/null; /root/repo/airlift_amd/bin/airlift-align -ax sr $o $REF $R > got.sam 2>/dev/null; echo "$o: $(diff got.sam exp.sam | grep -c '^<') differing"; done
