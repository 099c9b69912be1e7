/*
 * A small bounded queue of pointers between two threads of a drop-in command line (parser -> scorer,
 * scorer -> printer).  push() blocks while `cap` items wait; pop() returns NULL once the producer
 * closed the queue and it has drained.
 */
#ifndef AGX_PIPE_H
#define AGX_PIPE_H

#include <pthread.h>
#include <stddef.h>

#define AGX_PIPE_CAP 4

typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    void *slot[AGX_PIPE_CAP];
    int cap, count, closed;
} agx_pipe;

static void agx_pipe_init(agx_pipe *q, int cap)
{
    pthread_mutex_init(&q->mu, NULL);
    pthread_cond_init(&q->cv, NULL);
    q->cap = cap < 1 ? 1 : cap > AGX_PIPE_CAP ? AGX_PIPE_CAP : cap;
    q->count = 0;
    q->closed = 0;
}

static void agx_pipe_push(agx_pipe *q, void *item)
{
    pthread_mutex_lock(&q->mu);
    while (q->count == q->cap) pthread_cond_wait(&q->cv, &q->mu);
    q->slot[q->count++] = item;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
}

static void agx_pipe_close(agx_pipe *q)
{
    pthread_mutex_lock(&q->mu);
    q->closed = 1;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
}

static void *agx_pipe_pop(agx_pipe *q)
{
    pthread_mutex_lock(&q->mu);
    while (q->count == 0 && !q->closed) pthread_cond_wait(&q->cv, &q->mu);
    void *item = NULL;
    if (q->count) {
        item = q->slot[0];
        for (int k = 1; k < q->count; k++) q->slot[k - 1] = q->slot[k];
        q->count--;
        pthread_cond_broadcast(&q->cv);
    }
    pthread_mutex_unlock(&q->mu);
    return item;
}

/* "0,0,1" -> devices[]; returns the count (0 = variable unset or empty) */
static int agx_parse_devices(const char *s, int *devices, int max)
{
    int n = 0;
    while (s && *s && n < max) {
        char *end = NULL;
        long v = strtol(s, &end, 10);
        if (end == s) break;
        devices[n++] = (int)v;
        s = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    return n;
}

#endif
