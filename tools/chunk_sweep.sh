for k in 18 19 20 21 22 23; do PANDA_TIMING=0 python tools/tabled_bench.py $k 0 0 9 0 0,16,20,24,28,32,40,48,64 2>&1 | grep tables | sed 's/(.*built in [0-9.]*s)//; s/group= 0//' | cut -c1-110; done
