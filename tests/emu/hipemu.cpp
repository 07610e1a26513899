// tests/emu/hipemu.cpp - TEST INFRASTRUCTURE ONLY: the cooperative-fiber runtime behind tests/emu/hip/hip_runtime.h.
//
// A launch runs its workgroups on a few OS threads; the threads of ONE workgroup are fibers on one OS thread
// (hand-rolled x86-64 context switch), scheduled round-robin.  A fiber runs until it reaches a rendezvous
// (workgroup barrier or cross-lane operation); an operation completes when every LIVE thread of the group has
// arrived (threads that returned from the kernel no longer count - like terminated wavefronts on the GPU).
// If no fiber can run, the kernel has divergent lanes in a cross-lane operation or an unmatched barrier: abort
// with a message instead of hanging.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <pthread.h>
#include <sys/mman.h>

#include <atomic>
#include <deque>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif

extern thread_local char hipemu_lds_begin[64];  // tests/emu/build_emu.py links these two around the kernel objects
extern thread_local char hipemu_lds_end[64];

namespace hipemu {

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

namespace {
constexpr size_t STACK = 96 * 1024;
constexpr int MAXT = 1024;

enum State { READY = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };

struct Fiber {
  void* sp;
  void* asan_fake = nullptr;  // AddressSanitizer's fake-stack handle while the fiber is switched out
  const void* stack_lo = nullptr;
  Ctx ctx;
  int lin, lane, wave;
  State state;
  uint64_t wait_gen;
};

struct Wave {
  int live = 0, arrived = 0;
  uint64_t gen = 0;
  uint64_t slot[2][64];      // exchange buffers, double-buffered by generation parity
  double ma[2][64], mb[2][64];  // MFMA operand buffers
};

struct Block {
  Fiber fibers[MAXT];
  Wave waves[MAXT / 64];
  int nthreads = 0, live = 0, arrived = 0;
  uint64_t gen = 0;
  int orv[2] = {0, 0};
  void* sched_sp = nullptr;
  void* sched_fake = nullptr;
  const void* sched_lo = nullptr;  // bounds of the scheduler's (the OS thread's) stack, for AddressSanitizer
  size_t sched_size = 0;
  Fiber* cur = nullptr;
  const std::function<void()>* body = nullptr;
  char* stacks = nullptr;
};

thread_local Block* t_blk = nullptr;

// AddressSanitizer must be told about every stack switch (it tracks the bounds of the running stack)
inline void asan_leave(void** fake_save, const void* to_lo, size_t to_size) {
#ifdef HIPEMU_ASAN
  __sanitizer_start_switch_fiber(fake_save, to_lo, to_size);
#endif
}
inline void asan_enter(void* fake_saved) {
#ifdef HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(fake_saved, nullptr, nullptr);
#endif
}

void yield_to_scheduler(bool dying = false) {
  Block* b = t_blk;
  Fiber* f = b->cur;
  asan_leave(dying ? nullptr : &f->asan_fake, b->sched_lo, b->sched_size);
  hipemu_switch(&f->sp, b->sched_sp);
  asan_enter(f->asan_fake);
}

void release_wave_if_complete(Wave& w) {
  if (w.live > 0 && w.arrived == w.live) {
    w.arrived = 0;
    ++w.gen;
  }
}
void release_block_if_complete(Block* b) {
  if (b->live > 0 && b->arrived == b->live) {
    b->arrived = 0;
    ++b->gen;
    b->orv[b->gen & 1] = 0;  // the buffer of the NEXT barrier generation
  }
}

void fiber_main() {
  Block* b = t_blk;
  Fiber* f = b->cur;
  asan_enter(nullptr);
  (*b->body)();
  // thread leaves the kernel: it no longer takes part in any rendezvous
  f->state = DONE;
  Wave& w = b->waves[f->wave];
  --w.live;
  --b->live;
  release_wave_if_complete(w);
  release_block_if_complete(b);
  yield_to_scheduler(true);
  fprintf(stderr, "hipemu: finished fiber resumed\n");
  abort();
}

// rendezvous of the live lanes of the caller's wave; returns the generation the caller arrived in
uint64_t wave_arrive() {
  Block* b = t_blk;
  Fiber* f = b->cur;
  Wave& w = b->waves[f->wave];
  const uint64_t g = w.gen;
  ++w.arrived;
  if (w.arrived == w.live) {
    w.arrived = 0;
    ++w.gen;
  } else {
    f->state = WAIT_WAVE;
    f->wait_gen = g;
    yield_to_scheduler();
  }
  return g;
}

// LDS of the kernels = the `static thread_local` arrays of the two kernel translation units, which the build links between
// the two marker objects below: [hipemu_lds_begin, bgp_fill, bgp_linalg, hipemu_lds_end].  A worker thread is created per
// launch, so those arrays start as ZEROS for nearly every workgroup - kinder than the GPU, where LDS holds whatever the
// previous workgroup left.  HIPEMU_POISON=ff fills them with 0xFF (fp64 NaN, int -1) before every workgroup: a kernel that
// reads LDS it has not written shows up.
void poison_lds() {
  static const bool on = [] {
    const char* e = getenv("HIPEMU_POISON");
    return e && (strtol(e, nullptr, 16) & 0xff) == 0xff;
  }();
  if (!on) return;
  char* lo = hipemu_lds_begin + sizeof(hipemu_lds_begin);
  char* hi = hipemu_lds_end;
  if (hi < lo || hi - lo > (ptrdiff_t)(64 << 20)) {
    fprintf(stderr, "hipemu: LDS markers out of order (%p .. %p): link order changed?\n", (void*)lo, (void*)hi);
    abort();
  }
  static std::atomic<bool> said{false};
  if (!said.exchange(true)) fprintf(stderr, "hipemu: LDS poison on (%zu bytes of kernel LDS arrays per worker thread)\n", (size_t)(hi - lo));
  memset(lo, 0xFF, (size_t)(hi - lo));
}

void run_block(Block* b, dim3 grid, dim3 block, dim3 bid, const std::function<void()>& body) {
  poison_lds();
  const int nt = (int)(block.x * block.y * block.z);
  b->nthreads = b->live = nt;
  b->arrived = 0;
  b->gen = 0;
  b->orv[0] = b->orv[1] = 0;
  b->body = &body;
  const int nw = (nt + 63) / 64;
  for (int w = 0; w < nw; ++w) {
    b->waves[w].live = (w == nw - 1) ? nt - 64 * w : 64;
    b->waves[w].arrived = 0;
    b->waves[w].gen = 0;
  }
  for (int t = 0; t < nt; ++t) {
    Fiber& f = b->fibers[t];
    f.lin = t;
    f.lane = t & 63;
    f.wave = t >> 6;
    f.state = READY;
    f.wait_gen = 0;
    f.ctx.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    f.ctx.bid = bid;
    f.ctx.bdim = block;
    f.ctx.gdim = grid;
    char* top = b->stacks + (size_t)(t + 1) * STACK;  // 16-byte aligned
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;                                  // fake return address of fiber_main's "caller"
    *--sp = reinterpret_cast<void*>(&fiber_main);     // popped by `ret`
    for (int r = 0; r < 6; ++r) *--sp = nullptr;      // rbp rbx r12 r13 r14 r15
    f.sp = sp;
    f.stack_lo = top - STACK;
    f.asan_fake = nullptr;
  }
  t_blk = b;
#ifdef HIPEMU_ASAN
  if (b->sched_lo == nullptr) {
    pthread_attr_t attr;
    pthread_getattr_np(pthread_self(), &attr);
    void* lo = nullptr;
    size_t sz = 0;
    pthread_attr_getstack(&attr, &lo, &sz);
    pthread_attr_destroy(&attr);
    b->sched_lo = lo;
    b->sched_size = sz;
  }
#endif
  while (b->live > 0) {
    bool progressed = false;
    for (int t = 0; t < nt; ++t) {
      Fiber& f = b->fibers[t];
      if (f.state == DONE) continue;
      if (f.state == WAIT_WAVE && b->waves[f.wave].gen == f.wait_gen) continue;
      if (f.state == WAIT_BLOCK && b->gen == f.wait_gen) continue;
      f.state = READY;
      b->cur = &f;
      g_ctx = &f.ctx;
      asan_leave(&b->sched_fake, f.stack_lo, STACK);
      hipemu_switch(&b->sched_sp, f.sp);
      asan_enter(b->sched_fake);
      progressed = true;
    }
    if (!progressed) {
      int ww = 0, wb = 0;
      for (int t = 0; t < nt; ++t) {
        ww += b->fibers[t].state == WAIT_WAVE;
        wb += b->fibers[t].state == WAIT_BLOCK;
      }
      fprintf(stderr,
              "hipemu: deadlock in block (%u,%u,%u): %d threads live, %d waiting in a cross-lane operation, %d at a "
              "workgroup barrier - divergent lanes inside a wave-level operation or an unmatched __syncthreads\n",
              bid.x, bid.y, bid.z, b->live, ww, wb);
      abort();
    }
  }
  t_blk = nullptr;
  g_ctx = nullptr;
}

// one block context (fiber stacks) per OS thread, released when the thread ends: a launch runs its blocks on
// short-lived worker threads, and thousands of launches must not accumulate their stacks
struct BlockHolder {
  Block* b = nullptr;
  ~BlockHolder() {
    if (b) {
      if (b->stacks) munmap(b->stacks, STACK * MAXT);
      delete b;
    }
  }
};
Block* worker_block() {
  static thread_local BlockHolder h;
  if (!h.b) {
    h.b = new Block();
    void* m = mmap(nullptr, STACK * MAXT, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) {
      perror("hipemu: mmap of fiber stacks");
      abort();
    }
    h.b->stacks = static_cast<char*>(m);
  }
  return h.b;
}

int worker_count() {
  static int n = [] {
    const char* e = getenv("HIPEMU_THREADS");
    int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  return n;
}
}  // namespace

thread_local Ctx* g_ctx = nullptr;

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
  const int nt = (int)(block.x * block.y * block.z);
  if (nblocks == 0 || nt == 0) return;
  if (nt > MAXT) {
    fprintf(stderr, "hipemu: %d threads per workgroup\n", nt);
    abort();
  }
  if (g_ctx != nullptr) {
    fprintf(stderr, "hipemu: nested launch\n");
    abort();
  }
  auto work = [&](std::atomic<uint64_t>* next) {
    Block* b = worker_block();
    for (;;) {
      const uint64_t i = next->fetch_add(1);
      if (i >= nblocks) break;
      const dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((uint64_t)grid.x * grid.y)));
      run_block(b, grid, block, bid, body);
    }
  };
  std::atomic<uint64_t> next{0};
  const int nw = (int)std::min<uint64_t>(nblocks, (uint64_t)worker_count());
  if (nw <= 1) {
    work(&next);
    return;
  }
  std::vector<std::thread> pool;
  for (int w = 0; w < nw; ++w) pool.emplace_back(work, &next);
  for (auto& t : pool) t.join();
}

uint64_t wave_xchg(uint64_t mine, int srclane, int fallback_self) {
  Block* b = t_blk;
  Fiber* f = b->cur;
  Wave& w = b->waves[f->wave];
  const int buf = (int)(w.gen & 1);
  w.slot[buf][f->lane] = mine;
  wave_arrive();
  const int base = f->wave * 64;
  const bool src_ok = srclane >= 0 && srclane < 64 && base + srclane < b->nthreads;
  if (!src_ok) {
    if (fallback_self) return mine;
    fprintf(stderr, "hipemu: cross-lane read from lane %d outside the wave\n", srclane);
    abort();
  }
  return w.slot[buf][srclane];
}

int wave_first_lane() {
  Block* b = t_blk;
  Fiber* f = b->cur;
  for (int l = 0; l < 64; ++l) {
    const int t = f->wave * 64 + l;
    if (t < b->nthreads && b->fibers[t].state != DONE) return l;
  }
  return f->lane;
}

void wave_mfma_f64_16x16x4(double a, double bv, double* d4, int neg_a) {
  Block* b = t_blk;
  Fiber* f = b->cur;
  Wave& w = b->waves[f->wave];
  if (w.live != 64) {
    fprintf(stderr, "hipemu: MFMA with %d live lanes\n", w.live);
    abort();
  }
  const int buf = (int)(w.gen & 1);
  w.ma[buf][f->lane] = neg_a ? -a : a;
  w.mb[buf][f->lane] = bv;
  wave_arrive();
  const int n = f->lane & 15, mq = f->lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int m = mq + 4 * r;
    double acc = d4[r];
    for (int k = 0; k < 4; ++k) acc = __builtin_fma(w.ma[buf][m + 16 * k], w.mb[buf][n + 16 * k], acc);
    d4[r] = acc;
  }
}

void block_barrier() {
  Block* b = t_blk;
  Fiber* f = b->cur;
  const uint64_t g = b->gen;
  ++b->arrived;
  if (b->arrived == b->live) {
    b->arrived = 0;
    ++b->gen;
    b->orv[b->gen & 1] = 0;
  } else {
    f->state = WAIT_BLOCK;
    f->wait_gen = g;
    yield_to_scheduler();
  }
}

int block_or(int pred) {
  Block* b = t_blk;
  const int buf = (int)(b->gen & 1);
  if (pred) b->orv[buf] = 1;
  block_barrier();
  return b->orv[buf];
}

// ---- streams --------------------------------------------------------------------------------------------------
// HIPEMU_SCHED unset / "sync": every operation runs at once (host order).  Otherwise the streams are queues:
//   lazy          a synchronisation call runs the named stream (and whatever its event waits force), nothing else
//   eager         ... after first draining every OTHER stream as far as it can go
//   random[:seed] ... by running runnable queue heads in random order until the call is satisfied
//   prio:<p>      ... by always running the runnable head of the highest-priority stream; a stream's role is its
//                 creation index mod 4 (a handle creates main, panel, copy, bulk in that order) and <p> = 0..23
//                 picks the permutation of the four roles: each role gets its turn to run as far ahead as it can
// An operation therefore runs as LATE (lazy), or in as unexpected an order (eager, random), as the stream / event
// graph allows: two launches that are ordered only by luck on the GPU come out in the wrong order here.
namespace {
struct StreamQ;
struct Item {
  int kind;  // 0 run, 1 record, 2 wait
  std::function<void()> fn;
  hipEvent_t ev;
  StreamQ* dep;
  uint64_t dep_seq;
};
struct StreamQ {
  std::deque<Item> q;
  uint64_t done = 0, enq = 0;
  int role = 0;  // creation index mod 4
};
struct Sched {
  int mode = 0;  // 0 sync, 1 lazy, 2 eager, 3 random, 4 prio
  int rank[4] = {0, 1, 2, 3};  // prio: rank of a role (0 = runs first)
  size_t created = 0;
  std::mt19937_64 rng{12345};
  std::recursive_mutex mu;
  std::vector<StreamQ*> streams;
  Sched() { set(getenv("HIPEMU_SCHED")); }
  void set(const char* e) {
    mode = 0;
    if (!e || !*e || !strcmp(e, "sync")) return;
    if (!strcmp(e, "lazy")) mode = 1;
    else if (!strcmp(e, "eager")) mode = 2;
    else if (!strncmp(e, "random", 6)) {
      mode = 3;
      rng.seed(e[6] == ':' ? (uint64_t)atoll(e + 7) : 12345);
    } else if (!strncmp(e, "prio:", 5)) {
      mode = 4;
      int p = atoi(e + 5) % 24, roles[4] = {0, 1, 2, 3};
      for (int i = 0; i < 4; ++i) {  // p-th permutation (factorial number system): position i holds role pick
        int f = 1;
        for (int k = 2; k <= 3 - i; ++k) f *= k;
        const int idx = p / f;
        p %= f;
        rank[roles[idx]] = i;
        for (int k = idx; k < 3 - i; ++k) roles[k] = roles[k + 1];
      }
    } else {
      fprintf(stderr, "hipemu: unknown HIPEMU_SCHED=%s\n", e);
      abort();
    }
  }
  // mutation testing of the event graph: the drop-th hipStreamWaitEvent (counted from the last reset) is ignored
  long wait_count = 0, wait_drop = -1;
};
Sched& sched() {
  static Sched s;
  return s;
}
inline StreamQ* qof(hipStream_t s) { return s ? static_cast<StreamQ*>(s->queue) : nullptr; }

bool head_runnable(StreamQ* s) {
  if (s->q.empty()) return false;
  const Item& it = s->q.front();
  return it.kind != 2 || it.dep->done >= it.dep_seq;
}
void run_head(StreamQ* s) {
  Item it = std::move(s->q.front());
  s->q.pop_front();
  if (it.kind == 0) it.fn();
  else if (it.kind == 1 && it.ev) it.ev->t_ms = now_ms();
  ++s->done;
}
void advance(StreamQ* s, uint64_t upto, int depth = 0) {
  if (depth > 64) {
    fprintf(stderr, "hipemu: cyclic event wait between streams\n");
    abort();
  }
  while (s->done < upto) {
    const Item& it = s->q.front();
    if (it.kind == 2 && it.dep->done < it.dep_seq) advance(it.dep, it.dep_seq, depth + 1);
    run_head(s);
  }
}
// make `s` reach position `upto` under the configured policy
void reach(StreamQ* s, uint64_t upto) {
  Sched& S = sched();
  if (S.mode == 2) {
    for (StreamQ* t : S.streams)
      if (t != s) {
        // as far as t can go without forcing s past what it needs anyway: run until its head waits on something
        // that is not done, resolving waits on streams other than s
        while (!t->q.empty()) {
          const Item& it = t->q.front();
          if (it.kind == 2 && it.dep->done < it.dep_seq) {
            if (it.dep == s) break;
            advance(it.dep, it.dep_seq);
          }
          run_head(t);
        }
      }
    advance(s, upto);
  } else if (S.mode == 3) {
    while (s->done < upto) {
      std::vector<StreamQ*> ready;
      for (StreamQ* t : S.streams)
        if (head_runnable(t)) ready.push_back(t);
      if (ready.empty()) {
        fprintf(stderr, "hipemu: no runnable stream but a synchronisation is pending (cyclic waits?)\n");
        abort();
      }
      run_head(ready[S.rng() % ready.size()]);
    }
  } else if (S.mode == 4) {
    while (s->done < upto) {
      StreamQ* best = nullptr;
      for (StreamQ* t : S.streams)
        if (head_runnable(t) && (!best || S.rank[t->role] < S.rank[best->role])) best = t;
      if (!best) {
        fprintf(stderr, "hipemu: no runnable stream but a synchronisation is pending (cyclic waits?)\n");
        abort();
      }
      run_head(best);
    }
  } else {
    advance(s, upto);
  }
}
}  // namespace

bool deferred() { return sched().mode != 0; }

void stream_create(hipStream_t s) {
  Sched& S = sched();
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  StreamQ* q = new StreamQ();
  q->role = (int)(S.created++ % 4);
  s->queue = q;
  S.streams.push_back(q);
}
void stream_destroy(hipStream_t s) {
  Sched& S = sched();
  if (!s) return;
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  StreamQ* q = qof(s);
  reach(q, q->enq);
  // other queues may still hold waits on positions of this one: they are all done, keep the (empty) queue alive
  s->queue = nullptr;
}
void enqueue(hipStream_t s, std::function<void()> fn) {
  Sched& S = sched();
  if (!S.mode || !s) {  // immediate mode, or the null stream (which the non-blocking streams do not synchronise with)
    fn();
    return;
  }
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  StreamQ* q = qof(s);
  q->q.push_back(Item{0, std::move(fn), nullptr, nullptr, 0});
  ++q->enq;
}
void event_record(hipEvent_t e, hipStream_t s) {
  Sched& S = sched();
  if (!S.mode || !s) {
    e->t_ms = now_ms();
    e->rec_queue = nullptr;
    return;
  }
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  StreamQ* q = qof(s);
  // an earlier record of the same event that is still queued must not write through a stale pointer later: it stays
  // valid (the event object lives until hipEventDestroy, which flushes)
  q->q.push_back(Item{1, nullptr, e, nullptr, 0});
  e->rec_queue = q;
  e->rec_seq = ++q->enq;
}
void event_forget(hipEvent_t e) {
  Sched& S = sched();
  if (!S.mode || !e->rec_queue) return;
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  // every queued record of this event holds its pointer: flush all queues up to their current end
  for (StreamQ* t : S.streams)
    for (const Item& it : t->q)
      if (it.kind == 1 && it.ev == e) {
        reach(t, t->enq);
        break;
      }
}
void stream_wait_event(hipStream_t s, hipEvent_t e) {
  Sched& S = sched();
  if (!S.mode || !s || !e->rec_queue) return;  // never recorded (or recorded in immediate mode): nothing to wait for
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  StreamQ* q = qof(s);
  StreamQ* dep = static_cast<StreamQ*>(e->rec_queue);
  if (S.wait_count++ == S.wait_drop) return;  // mutation under test
  // the wait captures the record that is current NOW (a later re-record of the event does not move it)
  q->q.push_back(Item{2, nullptr, nullptr, dep, e->rec_seq});
  ++q->enq;
}
void stream_sync(hipStream_t s) {
  Sched& S = sched();
  if (!S.mode || !s) return;
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  StreamQ* q = qof(s);
  if (q) reach(q, q->enq);
}
void event_sync(hipEvent_t e) {
  Sched& S = sched();
  if (!S.mode || !e->rec_queue) return;
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  reach(static_cast<StreamQ*>(e->rec_queue), e->rec_seq);
}
bool event_done(hipEvent_t e) {
  Sched& S = sched();
  if (!S.mode || !e->rec_queue) return true;
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  return static_cast<StreamQ*>(e->rec_queue)->done >= e->rec_seq;
}
void device_sync() {
  Sched& S = sched();
  if (!S.mode) return;
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  // (reverse creation order: one more chance for an unordered pair to come out the wrong way round)
  for (size_t i = S.streams.size(); i-- > 0;) reach(S.streams[i], S.streams[i]->enq);
}

}  // namespace hipemu

// control surface for the tests (ctypes): switch the policy between runs, drop one wait, count them
extern "C" {
void hipemu_set_sched(const char* policy) {
  hipemu::device_sync();
  std::lock_guard<std::recursive_mutex> lk(hipemu::sched().mu);
  hipemu::sched().set(policy);
}
void hipemu_drop_wait(long k) {
  std::lock_guard<std::recursive_mutex> lk(hipemu::sched().mu);
  hipemu::sched().wait_count = 0;
  hipemu::sched().wait_drop = k;
}
long hipemu_wait_count(void) { return hipemu::sched().wait_count; }
// a host-side stand-in for "work that torch enqueues on this stream" runs behind everything the stream already holds
void hipemu_stream_sync(void* stream) { hipemu::stream_sync(static_cast<hipStream_t>(stream)); }
}
