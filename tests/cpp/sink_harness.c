/* sink_harness.c -- the host program's packet sink (adsbdec_amd/csrc/cli/sink.c) without a GPU: formats `count`
 * pseudo-random frames with the library's own formatter (csrc/format.c), sends them through the sink in the host
 * program's batches and writes the same bytes to `copy` for the test to compare.
 *   sink_harness <mode 0|1|2> <addr> <format 0|1|2> <count> <copy path> [pause_ms between batches] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "adsbdec_amd.h"
#include "../../adsbdec_amd/csrc/cli/sink.h"

int main(int argc, char **argv)
{
    if (argc < 6)
        return 2;
    const int mode = atoi(argv[1]), fmt = atoi(argv[3]);
    const long count = atol(argv[4]);
    const int pause_ms = argc > 6 ? atoi(argv[6]) : 0;
    sink s;
    sink_init(&s, mode, argv[2]);
    if (getenv("ADSB_CLI_RETRY_S"))
        s.retry_s = (unsigned)atoi(getenv("ADSB_CLI_RETRY_S"));
    if (sink_wait_peer(&s) != 0)
        return 255;
    FILE *copy = fopen(argv[5], "wb");
    if (!copy)
        return 2;
    static char batch[65536 + 256];
    size_t fill = 0;
    unsigned long packets = 0, sent_batches = 0, lost_batches = 0;
    unsigned long long x = 88172645463325252ull;
    for (long i = 0; i < count; i++) {
        adsb_frame f;
        memset(&f, 0, sizeof f);
        for (int k = 0; k < 14; k++) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            f.frame[k] = (i % 5 == 0 && k % 3 == 0) ? 0x1a : (uint8_t)x; /* Beast escapes its 0x1a bytes */
        }
        f.len = (i % 3 == 0) ? 7 : 14;
        f.ts = (uint64_t)i * 2405 + (i % 7 == 0 ? 0x1a : 0);
        f.pw = (uint32_t)(x >> 40);
        f.g = (uint64_t)i * 2400;
        fill += (size_t)adsb_format_frame(&f, fmt, batch + fill);
        packets++;
        if (fill >= 65536 || i + 1 == count) {
            const int rc = sink_write(&s, batch, fill, packets);
            if (rc < 0)
                return 255;
            if (rc == 0) {
                fwrite(batch, 1, fill, copy);
                sent_batches++;
            } else {
                lost_batches++;
            }
            fill = 0, packets = 0;
            if (pause_ms)
                usleep((useconds_t)pause_ms * 1000);
        }
    }
    fclose(copy);
    const int close_rc = sink_close(&s);
    if (close_rc != 0) {
        fprintf(stderr, "the last block could not be written\n");
        return 1; /* (the host program's status for a stdout that is gone) */
    }
    fprintf(stderr, "batches sent %lu, lost %lu, packets dropped %llu\n", sent_batches, lost_batches, s.dropped);
    return 0;
}
