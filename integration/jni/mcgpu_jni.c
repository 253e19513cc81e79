/* JNI side of integration/java/gpu/McGpu.java: every native of that class over the C ABI of include/mcgpu.h.
 * UNVERIFIED: the build image has no JDK and no jni.h.  Build: integration/README.md.
 * Errors become ru.ifmo.genetics.utils.tool.ExecutionFailedException, what Tool.run turns into a logged message and
 * System.exit(1) (itmo!/utils/tool/Tool.java:450-462). */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mcgpu.h"

static jfieldID handle_field;

static void throw_failed(JNIEnv *env, const char *msg)
{
    jclass cls = (*env)->FindClass(env, "ru/ifmo/genetics/utils/tool/ExecutionFailedException");
    if (!cls) { (*env)->ExceptionClear(env); cls = (*env)->FindClass(env, "java/lang/RuntimeException"); }
    (*env)->ThrowNew(env, cls, msg ? msg : "libmcgpu error");
}

static mc_ctx *ctx_of(JNIEnv *env, jobject self)
{
    if (!handle_field) handle_field = (*env)->GetFieldID(env, (*env)->GetObjectClass(env, self), "handle", "J");
    return (mc_ctx *)(intptr_t)(*env)->GetLongField(env, self, handle_field);
}

JNIEXPORT jlong JNICALL Java_gpu_McGpu_create(JNIEnv *env, jclass cls, jint k, jint key_mode, jint device, jlong capacity_hint)
{
    (void)cls;
    mc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.k = k;
    cfg.key_mode = key_mode;
    cfg.device = device;
    cfg.capacity_hint = (uint64_t)capacity_hint;
    mc_ctx *ctx = NULL;
    if (mc_create(&cfg, &ctx) != MC_OK) { throw_failed(env, mc_last_error(NULL)); return 0; }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_gpu_McGpu_close(JNIEnv *env, jobject self)
{
    mc_ctx *ctx = ctx_of(env, self);
    if (ctx) mc_destroy(ctx);
    (*env)->SetLongField(env, self, handle_field, 0);
}

JNIEXPORT void JNICALL Java_gpu_McGpu_setCoverageHint(JNIEnv *env, jobject self, jint min_cov)
{
    mc_ctx *ctx = ctx_of(env, self);
    if (mc_set_coverage_hint(ctx, min_cov) != MC_OK) throw_failed(env, mc_last_error(ctx));
}

JNIEXPORT void JNICALL Java_gpu_McGpu_addReadsPacked(JNIEnv *env, jobject self, jlongArray words, jlongArray offsets, jint n_reads)
{
    mc_ctx *ctx = ctx_of(env, self);
    /* (not Get/ReleasePrimitiveArrayCritical: the call uploads, launches kernels and waits for the stream -- seconds for a
     * large batch -- and a critical region must not block: it holds up the collector for every loader thread) */
    jlong *w = (*env)->GetLongArrayElements(env, words, 0);
    jlong *o = (*env)->GetLongArrayElements(env, offsets, 0);
    int rc = (w && o) ? mc_add_reads_packed(ctx, (const uint64_t *)w, (const uint64_t *)o, (uint64_t)n_reads) : MC_ENOMEM;
    if (o) (*env)->ReleaseLongArrayElements(env, offsets, o, JNI_ABORT);
    if (w) (*env)->ReleaseLongArrayElements(env, words, w, JNI_ABORT);
    if (rc != MC_OK) throw_failed(env, mc_last_error(ctx));
}

JNIEXPORT jlong JNICALL Java_gpu_McGpu_addReadsFile(JNIEnv *env, jobject self, jstring path)
{
    mc_ctx *ctx = ctx_of(env, self);
    const char *p = (*env)->GetStringUTFChars(env, path, 0);
    uint64_t n = 0;
    int rc = p ? mc_add_reads_file(ctx, p, &n) : MC_ENOMEM;
    if (p) (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != MC_OK) { throw_failed(env, mc_last_error(ctx)); return 0; }
    return (jlong)n;
}

JNIEXPORT jlong JNICALL Java_gpu_McGpu_finalizeCounts(JNIEnv *env, jobject self)
{
    mc_ctx *ctx = ctx_of(env, self);
    uint64_t n = 0;
    if (mc_finalize_counts(ctx, &n) != MC_OK) { throw_failed(env, mc_last_error(ctx)); return 0; }
    return (jlong)n;
}

JNIEXPORT jshortArray JNICALL Java_gpu_McGpu_get(JNIEnv *env, jobject self, jlongArray keys)
{
    mc_ctx *ctx = ctx_of(env, self);
    const jsize n = (*env)->GetArrayLength(env, keys);
    jshortArray out = (*env)->NewShortArray(env, n);
    if (!out) return NULL;
    jlong *k = (*env)->GetLongArrayElements(env, keys, 0);
    jshort *v = (jshort *)malloc((size_t)(n > 0 ? n : 1) * sizeof(jshort));
    int rc = (k && v) ? mc_get(ctx, (const int64_t *)k, (uint64_t)n, (int16_t *)v) : MC_ENOMEM;
    if (k) (*env)->ReleaseLongArrayElements(env, keys, k, JNI_ABORT);
    if (rc == MC_OK) (*env)->SetShortArrayRegion(env, out, 0, n, v);
    free(v);
    if (rc != MC_OK) { throw_failed(env, mc_last_error(ctx)); return NULL; }
    return out;
}

/* one BfsResult object from one mc_bfs_result (NULL when the job found no solid seed k-mer) */
static jobject make_result(JNIEnv *env, jclass rcls, const mc_bfs_result *r)
{
    if (r->n == 0) return NULL;
    jobject o = (*env)->AllocObject(env, rcls);
    if (!o) return NULL;
    const jsize n = (jsize)r->n;
    jlongArray hi = (*env)->NewLongArray(env, n), lo = (*env)->NewLongArray(env, n);
    jintArray dist = (*env)->NewIntArray(env, n);
    jshortArray cov = (*env)->NewShortArray(env, n);
    jbyteArray last = (*env)->NewByteArray(env, n);
    if (!hi || !lo || !dist || !cov || !last) return NULL;
    (*env)->SetLongArrayRegion(env, hi, 0, n, (const jlong *)r->hi);
    (*env)->SetLongArrayRegion(env, lo, 0, n, (const jlong *)r->lo);
    (*env)->SetIntArrayRegion(env, dist, 0, n, (const jint *)r->dist);
    (*env)->SetShortArrayRegion(env, cov, 0, n, (const jshort *)r->cov);
    (*env)->SetByteArrayRegion(env, last, 0, n, (const jbyte *)r->last);
    (*env)->SetObjectField(env, o, (*env)->GetFieldID(env, rcls, "hi", "[J"), hi);
    (*env)->SetObjectField(env, o, (*env)->GetFieldID(env, rcls, "lo", "[J"), lo);
    (*env)->SetObjectField(env, o, (*env)->GetFieldID(env, rcls, "dist", "[I"), dist);
    (*env)->SetObjectField(env, o, (*env)->GetFieldID(env, rcls, "cov", "[S"), cov);
    (*env)->SetObjectField(env, o, (*env)->GetFieldID(env, rcls, "last", "[B"), last);
    (*env)->SetLongField(env, o, (*env)->GetFieldID(env, rcls, "levels", "J"), (jlong)r->levels);
    (*env)->SetLongField(env, o, (*env)->GetFieldID(env, rcls, "lookups", "J"), (jlong)r->lookups);
    return o;
}

JNIEXPORT jobjectArray JNICALL Java_gpu_McGpu_bfsBatch(JNIEnv *env, jobject self, jobjectArray seed_hi, jobjectArray seed_lo, jintArray dir,
                                                       jint min_cov, jlong max_kmers, jlong max_radius)
{
    mc_ctx *ctx = ctx_of(env, self);
    const jsize nj = (*env)->GetArrayLength(env, dir);
    jclass rcls = (*env)->FindClass(env, "gpu/McGpu$BfsResult");
    jobjectArray out = rcls ? (*env)->NewObjectArray(env, nj, rcls, NULL) : NULL;
    if (!out) return NULL;
    mc_bfs_job *jobs = (mc_bfs_job *)calloc((size_t)(nj > 0 ? nj : 1), sizeof *jobs);
    mc_bfs_result *res = (mc_bfs_result *)calloc((size_t)(nj > 0 ? nj : 1), sizeof *res);
    jlongArray *ah = (jlongArray *)calloc((size_t)(nj > 0 ? nj : 1), sizeof *ah), *al = (jlongArray *)calloc((size_t)(nj > 0 ? nj : 1), sizeof *al);
    jint *d = (*env)->GetIntArrayElements(env, dir, 0);
    int rc = (jobs && res && ah && al && d) ? MC_OK : MC_ENOMEM;
    for (jsize j = 0; j < nj && rc == MC_OK; j++) {
        ah[j] = (jlongArray)(*env)->GetObjectArrayElement(env, seed_hi, j);
        al[j] = (jlongArray)(*env)->GetObjectArrayElement(env, seed_lo, j);
        jobs[j].n_seeds = (uint64_t)(*env)->GetArrayLength(env, al[j]);
        jobs[j].seed_hi = (const uint64_t *)(*env)->GetLongArrayElements(env, ah[j], 0);
        jobs[j].seed_lo = (const uint64_t *)(*env)->GetLongArrayElements(env, al[j], 0);
        jobs[j].dir = d[j];
        if (!jobs[j].seed_hi || !jobs[j].seed_lo) rc = MC_ENOMEM;
    }
    if (rc == MC_OK) rc = mc_bfs_batch(ctx, jobs, (uint32_t)nj, min_cov, (int64_t)max_kmers, (int64_t)max_radius, res);
    for (jsize j = 0; j < nj; j++) {
        if (jobs && jobs[j].seed_hi) (*env)->ReleaseLongArrayElements(env, ah[j], (jlong *)jobs[j].seed_hi, JNI_ABORT);
        if (jobs && jobs[j].seed_lo) (*env)->ReleaseLongArrayElements(env, al[j], (jlong *)jobs[j].seed_lo, JNI_ABORT);
    }
    if (d) (*env)->ReleaseIntArrayElements(env, dir, d, JNI_ABORT);
    if (rc == MC_OK) {
        int pending = 0;  /* a Java exception is pending: no further JNI calls that may throw, but every result block is freed */
        for (jsize j = 0; j < nj; j++) {
            if (!pending) {
                jobject o = make_result(env, rcls, &res[j]);
                pending = (*env)->ExceptionCheck(env);
                if (!pending) (*env)->SetObjectArrayElement(env, out, j, o);
            }
            mc_bfs_result_free(&res[j]);
        }
        if (pending) out = NULL;
    }
    free(jobs); free(res); free(ah); free(al);
    if (rc != MC_OK) { throw_failed(env, mc_last_error(ctx)); return NULL; }
    return out;
}
