/* sink.h -- where the host program's packets go: stdout, or one TCP peer (-s connect / -l listen), the reference's
 * three output modes (main.c:65-72 `outmode`, output.c:59-157 initNet, output.c:318-336 the writer loop's write side).
 * Part of the host program, not of the library's ABI. */
#ifndef ADSBDEC_AMD_CLI_SINK_H
#define ADSBDEC_AMD_CLI_SINK_H
#include <stddef.h>

enum { SINK_STDOUT = 0, SINK_CONNECT = 1, SINK_LISTEN = 2 };

typedef struct {
    int mode;
    int fd;              /* the connected peer, or -1 */
    int had_peer;        /* a peer has been connected at some point */
    const char *rawaddr; /* "host[:port]" or "[v6addr][:port]"; default port 30001 (-s) / 30002 (-l), output.c:84,93 */
    unsigned retry_s;    /* pause between attempts while no peer can be had (3 s, output.c:282); tests shorten it */
    unsigned long long dropped; /* packets given up after a peer went away (the reference frees its queue, output.c:325) */
    const volatile int *stop;   /* set by the program's SIGINT / SIGTERM / SIGQUIT handler (main.c:91-96): waiting for a peer ends */
} sink;

void sink_init(sink *s, int mode, const char *rawaddr);
/* One round of initNet: 0 = a peer is connected, 1 = none could be had this time (try again later), -1 = the address
 * cannot be used at all (message on stderr; the program ends with status 255 like runOutput() == -1).  Prints
 * "listening" / "connected" on stderr like the reference. */
int sink_establish(sink *s);
/* Blocks until a peer is there (sink_establish every retry_s seconds).  -1 only for an unusable address; 2: *stop was set
 * while waiting (the program was told to end: handlerExit, output.c:346-353). */
int sink_wait_peer(sink *s);
/* Write one batch of packets (any bytes).  stdout: 0 / -1 (stdout is gone).  TCP: the first call waits for a peer; a
 * peer that goes away gets "disconnected" on stderr and the rest of the batch is dropped (returns 1); every later batch
 * tries ONCE to connect again (-s) and is dropped without a peer; a listening sink (-l) does not listen again.
 * -1: the address is unusable. */
int sink_write(sink *s, const char *buf, size_t len, unsigned long packets);
/* Flushes stdout / closes the peer.  -1: the last buffered bytes could not be written (disk full, closed pipe). */
int sink_close(sink *s);
#endif
