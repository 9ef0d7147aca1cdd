/* oracle/ref_harness/serial_threads.c — TEST INFRASTRUCTURE ONLY.
 *
 * LD_PRELOAD shim that makes the reference `pagraph -t N` deterministic (SURVEY.md §8c "TN"):
 * every pthread_create runs the new thread to completion before returning, so the reference's
 * strided std::thread loops (PAGraph/src/tools/thread/MultiThreadTools.tcc:5-21) execute
 * thread-major: t = 0 handles items 0, N, 2N, ...; then t = 1; ...  The later join issued by
 * std::thread::join() must not join a second time, so joined ids are remembered.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stddef.h>

#define MAX_DONE 4096
static pthread_t done_ids[MAX_DONE];
static void *done_ret[MAX_DONE];
static int done_n = 0;
static pthread_mutex_t done_mu = PTHREAD_MUTEX_INITIALIZER;

typedef int (*create_fn)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
typedef int (*join_fn)(pthread_t, void **);

int pthread_create(pthread_t *th, const pthread_attr_t *attr, void *(*fn)(void *), void *arg) {
    static create_fn real_create = NULL;
    static join_fn real_join = NULL;
    if (!real_create) real_create = (create_fn)dlsym(RTLD_NEXT, "pthread_create");
    if (!real_join) real_join = (join_fn)dlsym(RTLD_NEXT, "pthread_join");
    int rc = real_create(th, attr, fn, arg);
    if (rc != 0) return rc;
    void *ret = NULL;
    real_join(*th, &ret);
    pthread_mutex_lock(&done_mu);
    if (done_n == MAX_DONE) done_n = 0; /* ring: ids are consumed by the matching join */
    done_ids[done_n] = *th;
    done_ret[done_n] = ret;
    ++done_n;
    pthread_mutex_unlock(&done_mu);
    return 0;
}

int pthread_join(pthread_t th, void **retval) {
    static join_fn real_join = NULL;
    if (!real_join) real_join = (join_fn)dlsym(RTLD_NEXT, "pthread_join");
    pthread_mutex_lock(&done_mu);
    for (int i = done_n - 1; i >= 0; --i) {
        if (pthread_equal(done_ids[i], th)) {
            if (retval) *retval = done_ret[i];
            done_ids[i] = done_ids[done_n - 1];
            done_ret[i] = done_ret[done_n - 1];
            --done_n;
            pthread_mutex_unlock(&done_mu);
            return 0;
        }
    }
    pthread_mutex_unlock(&done_mu);
    return real_join(th, retval);
}
