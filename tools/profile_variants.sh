cd $GRAFT_REPO_ROOT
TAG=${1:-r02_vXX}
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}.log 2>&1
O=gpurun_out/${TAG}
for a in "myelin bf16" "semseg_spine f16" "semseg_axon bf16" "mivcsj f16" "syntype bf16"; do set -- $a
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --arch $1 --act $2 > $O/bench_$1_$2.json 2>>$O/variants.err
done
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload config3 > $O/bench_config3.json 2>>$O/variants.err
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload config4 --geometry tile128 > $O/bench_config4_tile128.json 2>>$O/variants.err
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload config4 --geometry reference > $O/bench_config4_reference.json 2>>$O/variants.err
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload config5 --geometry tile128 > $O/bench_config5.json 2>>$O/variants.err
python3 tools/ref_geometry_check.py > $O/ref_geometry.txt 2>&1
tail -3 $O/ref_geometry.txt
