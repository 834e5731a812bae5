/* Plain-C use of the C-ABI (include/fdc_amd.h): the throughput chain of the FrequencyDomainChannelizer hier block
 * (python/FrequencyDomainChannelizer.py:200-231 of gr-FDC) for N = 4096, R = 4 and the example flowgraph's four
 * channels, fed with a tone in channel 0.  Build (no hipcc needed on the caller's side):
 *   gcc -std=c99 -O2 -Iinclude examples/fdc_pipeline_example.c -Lgr-fdc_amd -lfdc_amd -lm -Wl,-rpath,$PWD/gr-fdc_amd -o fdc_example
 * Prints the RMS amplitude of every channel's output: ~1 for the channel that holds the tone, ~0 elsewhere. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fdc_amd.h"

int main(void)
{
    enum { N = 4096, R = 4, NB = 8, C = 4 };
    const int H = N - N / R;
    /* (f, l, passbw, stopbw) as get_opt_channelparams derives them for [[0.12,0.05],[0.22,0.1],[-0.14,0.12],[0,0.081]] */
    const fdc_channel chans[C] = {{2412, 256, 0.88f, 1.0f}, {2693, 512, 0.88f, 1.0f}, {963, 1024, 0.528f, 0.778f}, {1792, 512, 0.7128f, 1.0f}};
    fdc_pipeline_cfg cfg = {0, N, R, FDC_WIN_HANN, C, chans, NB, 0, 0, 0, 0, 0};   /* flags 0: the library picks the kernels */
    fdc_pipeline *p = NULL;
    if (fdc_pipeline_create(&cfg, &p) != FDC_OK) { fprintf(stderr, "create: %s\n", fdc_last_error()); return 1; }

    float *x = (float *)malloc(sizeof(float) * 2 * (size_t)NB * H);
    const int k = (2412 + 128) - N / 2;                 /* unshifted bin at the centre of channel 0 */
    for (long n = 0; n < (long)NB * H; n++) {
        const double ph = 2.0 * 3.14159265358979323846 * (double)((k * n) % N) / N;
        x[2 * n] = (float)cos(ph); x[2 * n + 1] = (float)sin(ph);
    }
    void *outs[C];
    for (int c = 0; c < C; c++) outs[c] = malloc(sizeof(float) * 2 * (size_t)NB * fdc_pipeline_channel_lout(p, c));
    const int n = fdc_pipeline_work(p, x, NB, outs, NULL);
    if (n != NB) { fprintf(stderr, "work: %s\n", fdc_last_error()); return 1; }
    for (int c = 0; c < C; c++) {
        const int lout = fdc_pipeline_channel_lout(p, c);
        const float *y = (const float *)outs[c] + 2 * (size_t)lout;          /* skip the first block (zero history) */
        double acc = 0.0;
        for (long i = 0; i < (long)(NB - 1) * lout; i++) acc += (double)y[2 * i] * y[2 * i] + (double)y[2 * i + 1] * y[2 * i + 1];
        printf("channel %d: l=%d lout=%d rms=%.6f\n", c, chans[c].l, lout, sqrt(acc / ((NB - 1) * (double)lout)));
        free(outs[c]);
    }
    free(x);
    fdc_pipeline_destroy(p);
    return 0;
}
