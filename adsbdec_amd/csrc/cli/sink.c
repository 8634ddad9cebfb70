/* sink.c -- see sink.h.  Behaviour followed (not code): output.c:59-157 (address forms, default ports, one accepted
 * peer, the "listening" / "connected" / "disconnected" lines), output.c:277-285 (no peer: wait 3 s and try again),
 * output.c:318-331 (short writes are continued, a failed write drops what was queued), main.c:97-98 (SIGPIPE ignored:
 * here every send carries MSG_NOSIGNAL instead). */
#include "sink.h"

#include <netdb.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>

void sink_init(sink *s, int mode, const char *rawaddr)
{
    memset(s, 0, sizeof *s);
    s->mode = mode;
    s->fd = -1;
    s->rawaddr = rawaddr;
    s->retry_s = 3;
}

/* Split "host:port", "host", "[v6]:port", "[v6]" in place.  Returns 0, or -1 for "[..." without its bracket. */
static int split_address(char *text, int mode, int *family, char **host, const char **port)
{
    const char *dflt = mode == SINK_CONNECT ? "30001" : "30002";
    *family = AF_UNSPEC;
    *host = text;
    *port = dflt;
    if (text[0] == '[') {
        char *close_br = strchr(text + 1, ']');
        if (!close_br)
            return -1;
        *family = AF_INET6;
        *host = text + 1;
        *close_br = 0;
        if (close_br[1] == ':')
            *port = close_br + 2;
        return 0;
    }
    char *colon = strchr(text, ':'); /* the first one, like the reference: a bare v6 address needs its brackets */
    if (colon) {
        *colon = 0;
        *port = colon + 1;
    }
    return 0;
}

int sink_establish(sink *s)
{
    if (s->mode == SINK_STDOUT)
        return 0;
    if (!s->rawaddr)
        return -1;
    char *text = strdup(s->rawaddr);
    if (!text)
        return -1;
    int family;
    char *host;
    const char *port;
    if (split_address(text, s->mode, &family, &host, &port) != 0) {
        fprintf(stderr, "Invalid IPV6 address\n");
        free(text);
        return -1;
    }
    struct addrinfo hints, *list = NULL;
    memset(&hints, 0, sizeof hints);
    hints.ai_family = family;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host, port, &hints, &list) != 0) {
        fprintf(stderr, "Invalid/unknown address %s\n", host);
        free(text);
        return -1;
    }
    free(text);
    int got = -1;
    for (struct addrinfo *p = list; p && got < 0; p = p->ai_next) {
        const int sock = socket(p->ai_family, p->ai_socktype, p->ai_protocol);
        if (sock < 0)
            continue;
        if (s->mode == SINK_CONNECT) {
            if (connect(sock, p->ai_addr, p->ai_addrlen) == 0) {
                got = sock;
                fprintf(stderr, "connected\n");
            } else {
                close(sock);
            }
            continue;
        }
        const int one = 1; /* (not in the reference: lets a run rebind the port its predecessor just left) */
        setsockopt(sock, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (bind(sock, p->ai_addr, p->ai_addrlen) == 0 && listen(sock, 1) == 0) {
            fprintf(stderr, "listening\n");
            fflush(stderr);
            got = accept(sock, NULL, NULL); /* ONE peer; the listening socket goes away with it (output.c:139-146) */
            close(sock);                    /* ... BEFORE the word on stderr: whoever reads "connected" finds the port closed */
            if (got >= 0)
                fprintf(stderr, "connected\n");
            continue;
        }
        close(sock);
    }
    freeaddrinfo(list);
    fflush(stderr);
    s->fd = got;
    if (got >= 0)
        s->had_peer = 1;
    return got >= 0 ? 0 : 1;
}

int sink_wait_peer(sink *s)
{
    while (s->mode != SINK_STDOUT && s->fd < 0) {
        if (s->stop && *s->stop)
            return 2;
        const int rc = sink_establish(s); /* (a signal interrupts accept() / connect(): no SA_RESTART, like main.c:91-96) */
        if (rc < 0)
            return -1;
        if (rc > 0 && !(s->stop && *s->stop))
            sleep(s->retry_s);
    }
    return 0;
}

int sink_write(sink *s, const char *buf, size_t len, unsigned long packets)
{
    if (s->mode == SINK_STDOUT)
        return fwrite(buf, 1, len, stdout) == len ? 0 : -1;
    if (s->fd < 0) {
        /* Before the first peer: wait for one, however long (output.c:277-285).  Behind a peer that went away: the
         * reference, reading a file, is at its end by then and exits with its queue unsent (a listening one first sits
         * in accept() until somebody connects); here the input is decoded faster than a peer comes back, so a
         * connecting sink makes one attempt per batch and drops the batch without a peer, and a listening sink does
         * not listen again: the rest of the run's packets are dropped. */
        int rc;
        if (!s->had_peer) {
            rc = sink_wait_peer(s);
            if (rc == 2)
                rc = 1; /* told to end while waiting for the first peer: the batch goes nowhere */
        }
        else if (s->mode == SINK_CONNECT)
            rc = sink_establish(s);
        else
            rc = 1;
        if (rc < 0)
            return -1;
        if (rc > 0) {
            s->dropped += packets;
            return 1;
        }
    }
    while (len) {
        const ssize_t n = send(s->fd, buf, len, MSG_NOSIGNAL);
        if (n <= 0) {
            fprintf(stderr, "disconnected\n");
            fflush(stderr);
            close(s->fd);
            s->fd = -1;
            s->dropped += packets;
            return 1;
        }
        buf += n;
        len -= (size_t)n;
    }
    return 0;
}

int sink_close(sink *s)
{
    if (s->mode == SINK_STDOUT) /* the last block only reaches the file now: ENOSPC / EIO / EPIPE show up here */
        return (fflush(stdout) != 0 || ferror(stdout)) ? -1 : 0;
    if (s->fd >= 0) {
        close(s->fd);
        s->fd = -1;
    }
    return 0;
}
