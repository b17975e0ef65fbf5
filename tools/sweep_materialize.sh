# NCHW materialisation: block size x unroll variants (built by hand: make VARIANT="-DGYMHIP_MAT_BLOCK=b -DGYMHIP_MAT_UNR=u" OUT=../lib/libgymhip_m<b>_<u>.so), pieces per plane
for lib in "" _m256_1 _m256_8 _m512_4 _m512_2 _m1024_2 _m1024_1; do for sp in 1 2; do
  export GYMHIP_MAT_SPLIT=$sp; echo -n "lib=${lib:-default(256,4)} "
  GYMHIP_LIB=$PWD/gym_amd/lib/libgymhip$lib.so timeout -k 10 60 python tools/time_materialize.py 2>/dev/null | tr '\n' ' ' || exit 1; echo
done; done
