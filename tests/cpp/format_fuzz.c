/* format_fuzz.c -- adsb_format_frame (csrc/format.c == formatpkt, output.c:204-262) on random frames:
 * every packet length must stay inside the caller's 256-byte buffer, escape bytes (0x1a) and 48-bit
 * timestamp wrap included.  Built with -fsanitize=address,undefined by tests/test_host_logic.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "adsbdec_amd.h"

int main(void)
{
    srand(3);
    char pkt[256];
    unsigned long n = 0;
    for (int i = 0; i < 300000; i++) {
        adsb_frame f;
        memset(&f, 0, sizeof f);
        f.len = (rand() & 1) ? 7 : 14;
        for (int k = 0; k < 14; k++)
            f.frame[k] = (rand() % 5 == 0) ? 0x1a : (uint8_t)rand();
        f.ts = ((unsigned long long)rand() << 33) ^ ((unsigned long long)rand() << 11) ^ (unsigned long long)rand();
        if (i % 7 == 0)
            f.ts = 0x1a1a1a1a1a1aULL * 10 / 12; /* every timestamp byte needs escaping */
        f.pw = (unsigned)rand() * (unsigned)rand();
        for (int o = 0; o < 3; o++) {
            int l = adsb_format_frame(&f, o, pkt);
            const int max = o == 0 ? 1 + 28 + 2 : o == 1 ? 13 + 28 + 2 : 2 + 12 + 1 + 28;
            if (l <= 0 || l > max) {
                printf("outformat %d: length %d\n", o, l);
                return 1;
            }
            n += (unsigned long)l;
        }
    }
    printf("ok %lu\n", n);
    return 0;
}
