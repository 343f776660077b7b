// TEST INFRASTRUCTURE ONLY - fiber scheduler behind tests/hipemu/hip/hip_runtime.h.
// One OS thread; each work-item of a workgroup is a fiber with its own stack; fibers are resumed
// round-robin in linear-thread-id order and yield inside barriers / wave exchanges.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <vector>

namespace hipemu {

Ids cur;

// ---- minimal x86-64 SysV context switch (callee-saved registers only, no signal-mask syscalls) ----
extern "C" void hipemu_swap(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_swap
.type hipemu_swap,@function
hipemu_swap:
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
.size hipemu_swap,.-hipemu_swap
)");

static constexpr size_t kStack = 512 * 1024;
static constexpr int kMaxThreads = 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
};
struct Wave {
    int count = 0, gen = 0, alive = 0;
    alignas(16) unsigned char buf[64 * 128];
    size_t size = 0;
};

static Fiber fibers[kMaxThreads];
static Wave waves[kMaxThreads / 64];
static void* main_sp = nullptr;
static int cur_t = 0, n_threads = 0, alive = 0;
static int bar_count = 0, bar_gen = 0;
static long progress = 0;            // bumped whenever a barrier releases or a fiber finishes
static const std::function<void()>* body_ptr = nullptr;
static std::vector<unsigned char> smem;

static void yield_to_main() { hipemu_swap(&fibers[cur_t].sp, main_sp); }

static void release_if_complete() {
    if (bar_count > 0 && bar_count == alive) { bar_count = 0; ++bar_gen; ++progress; }
    for (int w = 0; w * 64 < n_threads; ++w) {
        Wave& wv = waves[w];
        if (wv.count > 0 && wv.count == wv.alive) { wv.count = 0; ++wv.gen; ++progress; }
    }
}

static void fiber_main() {
    (*body_ptr)();
    fibers[cur_t].done = true;
    --alive;
    --waves[cur_t / 64].alive;
    ++progress;
    release_if_complete();
    yield_to_main();
    fprintf(stderr, "hipemu: resumed a finished fiber\n");
    abort();
}

void syncthreads() {
    const int gen = bar_gen;
    ++bar_count;
    if (bar_count == alive) { bar_count = 0; ++bar_gen; ++progress; return; }
    while (bar_gen == gen) yield_to_main();
}

// __syncthreads_and: block-wide AND of a predicate (three rendezvous: deposit, read, re-arm)
static int and_acc = 1;
int syncthreads_and(int pred) {
    if (!pred) and_acc = 0;
    syncthreads();
    const int r = and_acc;
    syncthreads();
    if (cur_t == 0) and_acc = 1;
    syncthreads();
    return r;
}

static void wave_barrier(Wave& wv) {
    const int gen = wv.gen;
    ++wv.count;
    if (wv.count == wv.alive) { wv.count = 0; ++wv.gen; ++progress; return; }
    while (wv.gen == gen) yield_to_main();
}

int lane_id() { return cur_t & 63; }

void wave_sync() { wave_barrier(waves[cur_t / 64]); }

void wave_gather(const void* mine, void* all, size_t size) {
    if (size > 128) { fprintf(stderr, "hipemu: wave_gather payload too large\n"); abort(); }
    Wave& wv = waves[cur_t / 64];
    if (wv.count == 0) { memset(wv.buf, 0, sizeof(wv.buf)); wv.size = size; }
    if (wv.size != size) { fprintf(stderr, "hipemu: divergent wave op (payload %zu vs %zu)\n", wv.size, size); abort(); }
    memcpy(wv.buf + (size_t)lane_id() * size, mine, size);
    wave_barrier(wv);
    memcpy(all, wv.buf, 64 * size);
    wave_barrier(wv);      // nobody may overwrite buf before everyone has copied it
}

void* dyn_smem() { return smem.data(); }

static void ensure_stack(Fiber& f) {
    if (f.stack) return;
    void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    f.stack = static_cast<char*>(p);
}

void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    n_threads = (int)(block.x * block.y * block.z);
    if (n_threads <= 0 || n_threads > kMaxThreads) { fprintf(stderr, "hipemu: bad block size %d\n", n_threads); abort(); }
    smem.assign(shmem + 64, 0);
    body_ptr = &body;
    cur.bdim = block;
    cur.gdim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                alive = n_threads;
                bar_count = 0;
                for (int w = 0; w * 64 < n_threads; ++w) {
                    waves[w].count = 0;
                    waves[w].alive = (n_threads - w * 64) < 64 ? (n_threads - w * 64) : 64;
                }
                for (int t = 0; t < n_threads; ++t) {
                    Fiber& f = fibers[t];
                    ensure_stack(f);
                    f.done = false;
                    // initial frame: six callee-saved slots + return address = fiber_main; after the `ret`
                    // rsp must be 8 (mod 16), as at a normal function entry.
                    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
                    void** sp = reinterpret_cast<void**>(top - 16);
                    *--sp = nullptr;                                   // padding -> entry rsp = 8 mod 16
                    *--sp = reinterpret_cast<void*>(&fiber_main);      // return address
                    for (int i = 0; i < 6; ++i) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
                    f.sp = sp;
                }
                long last_progress = -1;
                int idle_rounds = 0;
                while (alive > 0) {
                    const long before = progress;
                    for (int t = 0; t < n_threads; ++t) {
                        if (fibers[t].done) continue;
                        cur_t = t;
                        cur.bid = dim3(bx, by, bz);
                        cur.tid = dim3((unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y)));
                        hipemu_swap(&main_sp, fibers[t].sp);
                    }
                    if (progress == before) {
                        if (++idle_rounds > 4) {
                            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d threads alive, barrier count %d "
                                            "(divergent __syncthreads or wave op?)\n", bx, by, bz, alive, bar_count);
                            abort();
                        }
                    } else {
                        idle_rounds = 0;
                    }
                    (void)last_progress;
                }
            }
    body_ptr = nullptr;
}

}  // namespace hipemu
