cd /root/repo
for st in 0 $((3*65536+64)) $((3*65536+32)) $((4*65536+64)) $((0*65536+64)) $((8*65536+64)) $((9*65536+64)) 0; do
  echo "== stagger bit $((st>>16)) amount $((st&65535))"
  DEVA_CONV_STAGGER=$st tools/convlab/convlab --libs tools/convlab/libconv_probes.so --iters 10 --shapes '512,512,1024,3,1,1,64,128,0,0,0;256,256,1024,3,1,1,64,128,0,0,0' | grep -v "^layer\|^frame\|^  big"
done
