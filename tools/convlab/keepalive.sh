cd /root/repo
SH='128,128,1024,3,1,1,64,128,0,0,0;512,512,1024,3,1,1,64,128,0,0,0;512,512,1024,3,1,4,64,128,0,0,0;256,0,1024,1,1,1,30,54,0,1,1'
for w in "0 1" "15 20" "40 40"; do
  set -- $w
  echo "== warm $1 ms time $2 ms"
  tools/convlab/convlab --libs tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so --iters 10 --warm_ms $1 --time_ms $2 --shapes "$SH" | grep -v "^layer\|^frame\|^  big"
done
