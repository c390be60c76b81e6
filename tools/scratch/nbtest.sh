cd tests/golden/g1_mt150pe && mkdir -p /tmp/g1 && for f in *.gz; do zcat $f > /tmp/g1/${f%.gz}; done; cd /tmp/g1
AL_DBG=$((1<<20)) /root/repo/airlift_amd/bin/airlift-align -ax sr MT-human.fa g1_1.fq g1_2.fq > out.sam 2> err.txt; echo "rc=$?"; head -30 err.txt
