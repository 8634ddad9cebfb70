/*
 * ref_chain_tail.c -- TEST INFRASTRUCTURE ONLY.
 *
 * oracle/Makefile compiles ONE translation unit made of, in this order,
 *     #include <stdint.h> / <string.h> / "adsbdec.h"      (what the slice needs)
 *     lines 29-101 of /root/reference/air.c, piped from where they lie
 *         (gain, fbuff/fidx, dsfilter, ampbuff/aidx, static decodeiq -- the part of
 *         air.c that touches no libairspy symbol; the Makefile asserts the slice
 *         boundaries and that it does not mention airspy)
 *     this file.
 * Nothing of the slice is written to disk or committed.  Being in the same
 * translation unit, the functions below can reach the reference's file-scope
 * statics and its static decodeiq; they add no behaviour of their own.
 */
void ref_decodeiq(const unsigned short *r, int len) { decodeiq(r, len); }   /* air.c:54 */
uint32_t ref_aidx(void) { return aidx; }                                     /* air.c:50 */
uint32_t ref_fidx(void) { return fidx; }                                     /* air.c:34 */
const float *ref_ampbuff(void) { return ampbuff; }                           /* air.c:49 */
