cd tests/golden/g6_repeats && mkdir -p /tmp/g6 && for f in *.gz; do zcat $f > /tmp/g6/${f%.gz}; done; cp meta.json /tmp/g6/; cd /root/repo
NORECHAIN=1 python tools/g6stage.py 2>&1 | tail -8
AL_DBG=134217728 python tools/g6stage.py 2>&1 | tail -8
