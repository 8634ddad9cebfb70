/*
 * format.c -- host-side frame formatter of libadsbdec_amd (plain C, no GPU).
 *
 * Produces the same bytes as formatpkt() in the reference (output.c:204-262,
 * built WITH_AIR as CMakeLists.txt:15 does) for its three output formats:
 *   0  AVR        '*' + uppercase hex + ";\n"                 (output.c:222-225,246-251)
 *   1  AVR-MLAT   '@' + 12 hex digits of the 12 MHz timestamp (output.c:226-229)
 *   2  Beast      0x1a, '2'|'3', 6-byte ts, level, frame, 0x1a doubled (output.c:230-243,253-259)
 * Quirks kept on purpose: the Beast level byte is not escaped and is
 * nearbyint(sqrt(pw))/8/5 in double precision truncated to a byte.
 */
#include <math.h>
#include <stdint.h>

#include "../../include/adsbdec_amd.h"

static int put_hex(char *p, const uint8_t *b, int n)
{
    static const char digits[] = "0123456789ABCDEF";
    for (int i = 0; i < n; i++) {
        *p++ = digits[b[i] >> 4];
        *p++ = digits[b[i] & 15];
    }
    return 2 * n;
}

int adsb_format_frame(const adsb_frame *f, int outformat, char *pkt)
{
    const uint64_t ts12 = (f->ts * 12) / 10; /* 10 MS/s loop count -> 12 MHz, output.c:218 */
    char *p = pkt;

    if (outformat == 2) {
        *p++ = 0x1a;
        *p++ = (f->len == 7) ? '2' : '3';
        for (int sh = 40; sh >= 0; sh -= 8) {
            const char ch = (char)(ts12 >> sh);
            *p++ = ch;
            if (ch == 0x1a)
                *p++ = ch;
        }
        *p++ = (char)(uint8_t)(nearbyint(sqrt((double)f->pw)) / 8 / 5);
        for (int i = 0; i < f->len; i++) {
            const char ch = (char)f->frame[i];
            *p++ = ch;
            if (ch == 0x1a)
                *p++ = ch;
        }
        return (int)(p - pkt);
    }

    if (outformat == 1) {
        uint8_t t[6];
        for (int i = 0; i < 6; i++)
            t[i] = (uint8_t)(ts12 >> (40 - 8 * i));
        *p++ = '@';
        p += put_hex(p, t, 6);
    } else {
        *p++ = '*';
    }
    p += put_hex(p, f->frame, f->len);
    *p++ = ';';
    *p++ = '\n';
    *p = 0;
    return (int)(p - pkt);
}
