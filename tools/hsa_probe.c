/* hsa_probe.c -- how much of a process's GPU start-up is the HSA runtime's own (hsa_init + finding the agent + one queue),
 * i.e. the floor under any host program on this platform, whatever it is written with.
 *   gcc -O2 tools/hsa_probe.c -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -Wl,-rpath,/opt/rocm/lib -o tools/bin/hsa_probe */
#include <hsa/hsa.h>
#include <stdio.h>
#include <time.h>
#include <unistd.h>

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static hsa_agent_t gpu;
static int have_gpu;
static hsa_status_t pick(hsa_agent_t a, void *data)
{
    hsa_device_type_t t;
    (void)data;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) {
        gpu = a;
        have_gpu = 1;
    }
    return HSA_STATUS_SUCCESS;
}
int main(void)
{
    double t0 = now_ms(), t = t0;
    hsa_status_t st = hsa_init();
    printf("%8.2f ms  hsa_init (status %d)\n", now_ms() - t, (int)st);
    t = now_ms();
    hsa_iterate_agents(pick, NULL);
    printf("%8.2f ms  hsa_iterate_agents (gpu found: %d)\n", now_ms() - t, have_gpu);
    t = now_ms();
    if (have_gpu) {
        hsa_queue_t *q = NULL;
        st = hsa_queue_create(gpu, 4096, HSA_QUEUE_TYPE_MULTI, NULL, NULL, 0xFFFFFFFFu, 0xFFFFFFFFu, &q);
        printf("%8.2f ms  hsa_queue_create #1 (status %d)\n", now_ms() - t, (int)st);
        t = now_ms();
        st = hsa_queue_create(gpu, 4096, HSA_QUEUE_TYPE_MULTI, NULL, NULL, 0xFFFFFFFFu, 0xFFFFFFFFu, &q);
        printf("%8.2f ms  hsa_queue_create #2 (status %d)\n", now_ms() - t, (int)st);
    }
    printf("%8.2f ms  total\n", now_ms() - t0);
    fflush(stdout);
    _exit(0);
}
