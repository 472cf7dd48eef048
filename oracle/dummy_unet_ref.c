/* ORACLE (test infrastructure, never linked into the product library).
 *
 * Plain-C fp32 restatement of the reference's simulator-path model and step loop:
 *   DummyUNet.forward        /root/reference/src/models/dummy_unet.py:37-59
 *       out = x + tanh(step/10) * Conv3d(SiLU(Conv3d(x))) + LayerNorm_C(x)
 *       Conv3d: kernel 3x3x3, stride 1, zero padding 1, bias      (:27-31)
 *       LayerNorm over the channel axis, biased variance, eps, affine (:33, :44-58)
 *   PipelineStage._run_local_steps   /root/reference/src/pipeline/pipeline.py:86-98
 *       for step in timesteps[start:end]: latent = model(latent, step)   (timestep VALUE passed)
 *
 * Pinned by tests/golden/dummy_*.npz, which were produced by importing the reference itself
 * (tests/golden/make_golden.py).  Accumulation order differs from MKL-DNN, so agreement is to
 * fp32 round-off (tests state 2e-5 relative), not bitwise.
 *
 * Layout: (B, C, F, H, W) contiguous float32, weights (Cout, Cin, 3, 3, 3) like torch.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void conv3d_3x3x3(const float *x, const float *w, const float *b, float *y,
                         int B, int Cin, int Cout, int F, int H, int W, int silu)
{
    const long plane = (long)H * W, vol = (long)F * plane;
    for (int n = 0; n < B; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *yo = y + ((long)n * Cout + co) * vol;
            for (long i = 0; i < vol; ++i) yo[i] = b ? b[co] : 0.0f;
            for (int ci = 0; ci < Cin; ++ci) {
                const float *xi = x + ((long)n * Cin + ci) * vol;
                const float *wk = w + ((long)co * Cin + ci) * 27;
                for (int kf = 0; kf < 3; ++kf)
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wv = wk[(kf * 3 + kh) * 3 + kw];
                            const int f0 = kf == 0 ? 1 : 0, f1 = kf == 2 ? F - 1 : F;
                            const int h0 = kh == 0 ? 1 : 0, h1 = kh == 2 ? H - 1 : H;
                            const int w0 = kw == 0 ? 1 : 0, w1 = kw == 2 ? W - 1 : W;
                            for (int f = f0; f < f1; ++f)
                                for (int h = h0; h < h1; ++h) {
                                    float *yr = yo + f * plane + (long)h * W;
                                    const float *xr = xi + (f + kf - 1) * plane + (long)(h + kh - 1) * W + (kw - 1);
                                    for (int ww = w0; ww < w1; ++ww) yr[ww] += wv * xr[ww];
                                }
                        }
            }
            if (silu)
                for (long i = 0; i < vol; ++i) yo[i] = yo[i] / (1.0f + expf(-yo[i]));
        }
}

/* one DummyUNet.forward; scratch must hold B*hidden*F*H*W floats */
void dummy_unet_forward_ref(const float *x, float *out, float *scratch,
                            const float *w1, const float *b1, const float *w2, const float *b2,
                            const float *ln_w, const float *ln_b, float ln_eps, int use_ln,
                            int B, int C, int hidden, int F, int H, int W, double step)
{
    const long vol = (long)F * H * W;
    const float gain = (float)tanh(step / 10.0);
    conv3d_3x3x3(x, w1, b1, scratch, B, C, hidden, F, H, W, 1);
    conv3d_3x3x3(scratch, w2, b2, out, B, hidden, C, F, H, W, 0);
    for (int n = 0; n < B; ++n)
        for (long p = 0; p < vol; ++p) {
            const float *xp = x + (long)n * C * vol + p;
            float *op = out + (long)n * C * vol + p;
            float mean = 0.f, var = 0.f;
            if (use_ln) {
                for (int c = 0; c < C; ++c) mean += xp[c * vol];
                mean /= (float)C;
                for (int c = 0; c < C; ++c) { float d = xp[c * vol] - mean; var += d * d; }
                var /= (float)C;
            }
            const float rstd = 1.0f / sqrtf(var + ln_eps);
            for (int c = 0; c < C; ++c) {
                float v = xp[c * vol] + gain * op[c * vol];
                if (use_ln) v += (xp[c * vol] - mean) * rstd * ln_w[c] + ln_b[c];
                op[c * vol] = v;
            }
        }
}

/* PipelineStage._run_local_steps over timesteps[start:end]; latent updated in place */
void dummy_pipeline_run_ref(float *latent, const int *timesteps, int start, int end,
                            const float *w1, const float *b1, const float *w2, const float *b2,
                            const float *ln_w, const float *ln_b, float ln_eps, int use_ln,
                            int B, int C, int hidden, int F, int H, int W)
{
    const long n = (long)B * C * F * H * W;
    float *tmp = (float *)malloc(sizeof(float) * n);
    float *scratch = (float *)malloc(sizeof(float) * (long)B * hidden * F * H * W);
    for (int i = start; i < end; ++i) {
        dummy_unet_forward_ref(latent, tmp, scratch, w1, b1, w2, b2, ln_w, ln_b, ln_eps, use_ln,
                               B, C, hidden, F, H, W, (double)timesteps[i]);
        memcpy(latent, tmp, sizeof(float) * n);
    }
    free(tmp);
    free(scratch);
}
