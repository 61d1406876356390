/* Test shim (not product code): calls the plain C-ABI entry points vmlmf_seq_forward / vmlmf_seq_backward behind a function that
 * has just filled 32 KB of stack with 0xA5, so that any field of a stack struct the library forgets to initialise is a wild
 * pointer rather than a lucky zero (vmlmf_extra.drop in the *_packed wrappers). */
#include <stddef.h>
#include <string.h>

typedef int (*fwd_t)(const void *, const void *, const float *, const float *, const float *, float *, float *, float *, void *,
                     void *, size_t, void *);
typedef int (*bwd_t)(const void *, const void *, const float *, const float *, const float *, const float *, const void *,
                     const float *, const float *, const float *, float *, float *, float *, const void *, void *, size_t, void *);

static __attribute__((noinline)) unsigned dirty(void) {
  volatile unsigned char buf[32768];
  memset((void *)buf, 0xA5, sizeof(buf));
  return buf[17] + buf[32000];
}

int dirty_forward(fwd_t f, const void *d, const void *p, const float *x, const float *h0, const float *c0, float *y, float *hT,
                  float *cT, void *reserve, void *ws, size_t wsb, void *stream) {
  if (dirty() == 0) return -1000;
  return f(d, p, x, h0, c0, y, hT, cT, reserve, ws, wsb, stream);
}

int dirty_backward(bwd_t f, const void *d, const void *p, const float *x, const float *h0, const float *c0, const float *y,
                   const void *reserve, const float *dy, const float *dhT, const float *dcT, float *dx, float *dh0, float *dc0,
                   const void *g, void *ws, size_t wsb, void *stream) {
  if (dirty() == 0) return -1000;
  return f(d, p, x, h0, c0, y, reserve, dy, dhT, dcT, dx, dh0, dc0, g, ws, wsb, stream);
}
