// tools/mfma4_layout_probe.cpp -- diagnostic: lane layout of v_mfma_f64_4x4x4_4b_f64 operands / result and the effect of
// the cbsz / abid (A-block broadcast) modifiers.  For every B lane p a one-hot B is multiplied with A[l] = 100 + l; the
// non-zero result lanes tell which (result lane, A lane) pairs meet B lane p.
// hipcc -O3 --offload-arch=gfx950 -o tools/mfma4_layout_probe tools/mfma4_layout_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CBSZ, int ABID>
__global__ void __launch_bounds__(64) probe(double *out)
{
    const int lane = threadIdx.x;
    const double a = 100.0 + lane;
    for (int p = 0; p < 64; p++) {
        const double b = (lane == p) ? 1.0 : 0.0;
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
        out[p * 64 + lane] = d;
    }
}

template <int CBSZ, int ABID>
void run(double *dev)
{
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dev);
    std::vector<double> h(64 * 64);
    hipMemcpy(h.data(), dev, sizeof(double) * 64 * 64, hipMemcpyDeviceToHost);
    printf("== cbsz=%d abid=%d: B lane p -> (result lane : A lane)\n", CBSZ, ABID);
    for (int p = 0; p < 64; p++) {
        printf("p=%2d:", p);
        for (int l = 0; l < 64; l++)
            if (h[p * 64 + l] != 0.0) printf(" %d:%d", l, (int)(h[p * 64 + l] - 100.0));
        printf("\n");
    }
}

int main()
{
    double *dev;
    hipMalloc(&dev, sizeof(double) * 64 * 64);
    run<0, 0>(dev);
    run<1, 0>(dev);
    run<1, 1>(dev);
    run<2, 0>(dev);
    run<2, 3>(dev);
    return 0;
}
