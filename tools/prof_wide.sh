# kernel statistics of a dim = 256 training step: which kernels do the GEMMs (run on the GPU box; output gpurun_out/r04_wide_model_kernel_stats.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf /tmp/p_wide
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_wide -- python $R/tools/wide_model_step.py 256 10 > /tmp/p_wide.log 2>&1
f=$(find /tmp/p_wide -name '*kernel_stats.csv' | head -1)
(grep 'ms per step' /tmp/p_wide.log
 echo "library GEMM kernels in the trace (Cijk_ / rocblas / hipblaslt / gemm in the name, other than dense_gemm_kernel):"
 (grep -i -E 'Cijk|rocblas|hipblas|gemm' $f | grep -v dense_gemm_kernel || echo "  none")
 echo "top kernels by total time:"
 head -25 $f | cut -c1-200) > $O/r04_wide_model_kernel_stats.txt
cat $O/r04_wide_model_kernel_stats.txt
