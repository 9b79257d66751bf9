// heapguard: an LD_PRELOAD page-guard allocator for hunting host heap corruption in a process that
// also runs PyTorch / HIP (no valgrind or host ASan build of those libraries in this image).
//
//   gcc -O2 -g -fPIC -shared -o tools/bin/libheapguard.so tools/heapguard.c -ldl -lpthread
//   LD_PRELOAD=tools/bin/libheapguard.so HEAPGUARD_MIN=32 HEAPGUARD_MAX=8192 python script.py
//   (the script calls heapguard_enable() through ctypes once start-up is over)
//
// Once enabled, every allocation whose size lies in [HEAPGUARD_MIN, HEAPGUARD_MAX] is placed at the END
// of its own pages inside a huge reserved arena, followed by an inaccessible guard page, and is never
// reused: free() makes its pages inaccessible for good.  A write past the end of a block or into a freed
// block therefore faults AT THE OFFENDING INSTRUCTION; the SIGSEGV handler prints the faulting thread's
// backtrace, the block's size and who freed it, then chains to the previous handler (Python's
// faulthandler, if enabled, adds the Python stacks).  Everything else goes to glibc's allocator.
// Debugging aid only: nothing in the library, the tests or the bench links or loads this.
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <execinfo.h>
#include <signal.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

extern void* __libc_malloc(size_t);
extern void __libc_free(void*);
extern void* __libc_calloc(size_t, size_t);
extern void* __libc_realloc(void*, size_t);
extern void* __libc_memalign(size_t, size_t);

#define PAGE 4096UL
#define ARENA_BYTES (1UL << 42)  // 4 TiB of address space, touched lazily
#define FRAMES 10

static char* g_base;                 // arena
static _Atomic uint64_t g_next;      // bump offset (bytes)
static uint32_t* g_size;             // per page: size of the block whose FIRST data page this is (0: none)
static uint8_t* g_state;             // per page: 1 live, 2 freed
static uint16_t* g_uoff;             // per page: offset of the user pointer inside the block's first page
static void** g_freed_by;            // per page: FRAMES return addresses of the free() call
static _Atomic int g_on;
static size_t g_min = 32, g_max = 8192;
static _Atomic uint64_t g_count, g_fallback;
static size_t (*real_usable)(void*);
static struct sigaction g_prev;

static inline int in_arena(const void* p) { return g_base && (const char*)p >= g_base && (const char*)p < g_base + ARENA_BYTES; }

static void* guard_alloc(size_t size, size_t align) {
  if (align < 16) align = 16;
  size_t padded = (size + align - 1) & ~(align - 1);
  size_t data = (padded + PAGE - 1) & ~(PAGE - 1);
  uint64_t off = atomic_fetch_add(&g_next, data + PAGE);  // data pages + one guard page
  if (off + data + PAGE > ARENA_BYTES) return NULL;
  char* first = g_base + off;
  if (mprotect(first, data, PROT_READ | PROT_WRITE) != 0) {  // VMA limit reached: let glibc serve it
    atomic_fetch_add(&g_fallback, 1);
    return NULL;
  }
  char* user = first + data - padded;
  uint64_t pg = off / PAGE;
  g_size[pg] = (uint32_t)size;
  g_uoff[pg] = (uint16_t)(user - first);
  g_state[pg] = 1;
  atomic_fetch_add(&g_count, 1);
  return user;
}

static void guard_free(void* p) {
  uint64_t pg = (uint64_t)((char*)p - g_base) / PAGE;
  if (g_state[pg] != 1) {
    fprintf(stderr, "heapguard: free(%p) of a block in state %d (double free?)\n", p, g_state[pg]);
    void* bt[32];
    int n = backtrace(bt, 32);
    backtrace_symbols_fd(bt, n, 2);
    abort();
  }
  size_t size = g_size[pg];
  size_t data = ((((uintptr_t)p & (PAGE - 1)) + size) + PAGE - 1) & ~(PAGE - 1);
  g_state[pg] = 2;
  void* bt[FRAMES + 2];
  int n = backtrace(bt, FRAMES + 2);
  for (int i = 0; i < FRAMES; ++i) g_freed_by[pg * FRAMES + i] = (i + 2 < n) ? bt[i + 2] : NULL;
  char* first = g_base + pg * PAGE;
  madvise(first, data, MADV_DONTNEED);
  mprotect(first, data, PROT_NONE);
}

static size_t guard_usable(void* p) {
  uint64_t pg = (uint64_t)((char*)p - g_base) / PAGE;
  return g_size[pg];
}

static void on_segv(int sig, siginfo_t* si, void* ctx) {
  char* a = (char*)si->si_addr;
  if (in_arena(a)) {
    static _Atomic int once;
    if (!atomic_exchange(&once, 1)) {
      uint64_t pg = (uint64_t)(a - g_base) / PAGE, q = pg;
      while (q > 0 && g_state[q] == 0 && pg - q < 64) --q;  // first data page of the block at / before the fault
      fprintf(stderr, "\nheapguard: invalid access at %p: page %lu, nearest block starts %lu page(s) before: size %u, %s\n",
              (void*)a, (unsigned long)pg, (unsigned long)(pg - q), g_size[q],
              g_state[q] == 2 ? "FREED (use after free)" : g_state[q] == 1 ? "live (overflow into its guard page)" : "?");
      fprintf(stderr, "heapguard: offset of the access from the block's start: %ld\n",
              (long)(a - (g_base + q * PAGE + g_uoff[q])));
      fprintf(stderr, "heapguard: faulting thread:\n");
      void* bt[48];
      int n = backtrace(bt, 48);
      backtrace_symbols_fd(bt, n, 2);
      if (g_state[q] == 2) {
        fprintf(stderr, "heapguard: the block was freed by:\n");
        int m = 0;
        while (m < FRAMES && g_freed_by[q * FRAMES + m]) ++m;
        backtrace_symbols_fd(&g_freed_by[q * FRAMES], m, 2);
      }
      fprintf(stderr, "heapguard: %lu guarded allocations so far, %lu fell back to glibc\n", (unsigned long)g_count,
              (unsigned long)g_fallback);
    }
  }
  // chain (faulthandler prints the Python stacks and re-raises)
  if (g_prev.sa_flags & SA_SIGINFO) {
    if (g_prev.sa_sigaction) { g_prev.sa_sigaction(sig, si, ctx); return; }
  } else if (g_prev.sa_handler != SIG_DFL && g_prev.sa_handler != SIG_IGN) {
    g_prev.sa_handler(sig);
    return;
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

static void* reserve(size_t bytes) {
  void* p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  return p == MAP_FAILED ? NULL : p;
}

int heapguard_enable(void) {
  if (g_on) return 0;
  const char* e;
  if ((e = getenv("HEAPGUARD_MIN"))) g_min = strtoul(e, NULL, 0);
  if ((e = getenv("HEAPGUARD_MAX"))) g_max = strtoul(e, NULL, 0);
  real_usable = (size_t(*)(void*))dlsym(RTLD_NEXT, "malloc_usable_size");
  void* a = mmap(NULL, ARENA_BYTES, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (a == MAP_FAILED) { perror("heapguard: arena"); return -1; }
  uint64_t pages = ARENA_BYTES / PAGE;
  g_size = reserve(pages * sizeof(uint32_t));
  g_state = reserve(pages);
  g_uoff = reserve(pages * sizeof(uint16_t));
  g_freed_by = reserve(pages * FRAMES * sizeof(void*));
  if (!g_size || !g_state || !g_freed_by) { perror("heapguard: tables"); return -1; }
  void* warm[4];
  backtrace(warm, 4);  // loads libgcc's unwinder now, not inside free()
  g_base = a;
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, &g_prev);
  atomic_store(&g_on, 1);
  fprintf(stderr, "heapguard: guarding allocations of %zu..%zu bytes\n", g_min, g_max);
  return 0;
}

void heapguard_report(void) {
  fprintf(stderr, "heapguard: %lu guarded allocations, %lu fell back to glibc, %.1f GiB of arena used\n",
          (unsigned long)g_count, (unsigned long)g_fallback, (double)g_next / (1UL << 30));
}

static inline int guarded(size_t n) { return g_on && n >= g_min && n <= g_max; }

void* malloc(size_t n) {
  if (guarded(n)) {
    void* p = guard_alloc(n, 16);
    if (p) return p;
  }
  return __libc_malloc(n);
}

void free(void* p) {
  if (!p) return;
  if (in_arena(p)) guard_free(p);
  else __libc_free(p);
}

void* calloc(size_t a, size_t b) {
  size_t n;
  if (__builtin_mul_overflow(a, b, &n)) { errno = ENOMEM; return NULL; }
  if (guarded(n)) {
    void* p = guard_alloc(n, 16);  // fresh pages: already zero
    if (p) return p;
  }
  return __libc_calloc(a, b);
}

void* realloc(void* p, size_t n) {
  if (!p) return malloc(n);
  if (n == 0) { free(p); return NULL; }
  if (!in_arena(p) && !guarded(n)) return __libc_realloc(p, n);
  size_t old = in_arena(p) ? guard_usable(p) : (real_usable ? real_usable(p) : n);
  void* q = malloc(n);
  if (!q) return NULL;
  memcpy(q, p, old < n ? old : n);
  free(p);
  return q;
}

void* memalign(size_t al, size_t n) {
  if (guarded(n) && al <= PAGE) {
    void* p = guard_alloc(n, al);
    if (p) return p;
  }
  return __libc_memalign(al, n);
}

void* aligned_alloc(size_t al, size_t n) { return memalign(al, n); }

int posix_memalign(void** out, size_t al, size_t n) {
  void* p = memalign(al, n);
  if (!p) return ENOMEM;
  *out = p;
  return 0;
}

size_t malloc_usable_size(void* p) {
  if (!p) return 0;
  if (in_arena(p)) return guard_usable(p);
  if (!real_usable) real_usable = (size_t(*)(void*))dlsym(RTLD_NEXT, "malloc_usable_size");
  return real_usable ? real_usable(p) : 0;
}
