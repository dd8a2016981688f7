// rc_dispatch.h -- the `rcorrector` CLI between start-up and exit: the one-pass ingest of a run without -c (files read once,
// their bases kept in HBM by the k-mer counter), the batch buffers, and the pipeline reader -> workers (one slot of a GPU's
// context each, rc_submit* / rc_wait*) -> writer, batches dealt to whichever GPU is free.
//   batching     main.cpp:439-523           batches never span files; mates travel together
//   one Store, T workers                    main.cpp:294-308,451,479-483
#pragma once
#include <atomic>
#include <memory>

#include "rc_writer.h"

// ---- k-mer counting pass (only without -c); one pass over the input where it fits: what the counting pass read stays ----
// The reference's pipeline reads every file twice -- jellyfish counts the k-mers (run_rcorrector.pl:262-281), stage 3
// corrects -- and so does the counting pass above followed by the correction loop.  When the inputs are plain files that
// fit (text in host memory, bases in HBM), the counting pass cuts them into the correction loop's batches right away:
// the text and its line index stay here, the sequence arenas stay in HBM with the counter (rc_table_count_keep), and the
// loop corrects them where they lie (rc_submit_resident): files are read, parsed and uploaded once.
struct Retained {
    int file = 0, mode = 0;
    bool fastq = true;
    int lpr_a = 4, lpr_b = 4;
    Block a, b;
    std::vector<uint32_t> off_a, off_b;
    int arena_a = 0, arena_b = 0;  // kept arenas of GPU `gpu`'s context
    int gpu = 0;
};

// everything the stages of a run share
struct Run {
    int k = 23, gpus = 1, inflight = 2, nworkers = 2;
    size_t batch_reads = (size_t)1 << 20, max_in_flight = 4;
    bool resident = false;  // one pass: the batches are `kept`, their bases in HBM
    bool numa_on = true, shared_gpu = false;
    char bad_q = 0;
    std::vector<ReadFile> files, mates;
    std::vector<rc_ctx *> ctx;                // one per GPU (the table is replicated)
    std::unique_ptr<std::mutex[]> submit_mu;  // rc_submit calls on one context are serialised
    std::vector<std::unique_ptr<Retained>> kept;
    std::vector<std::shared_ptr<Job>> warm_jobs;
    // pipeline state (guarded by mu)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Job>> order;  // submission order, for the writer
    std::vector<std::shared_ptr<Job>> pool;  // finished jobs: their buffers are reused (no fresh page faults)
    std::deque<std::shared_ptr<Job>> q;      // one queue for all workers: whichever context is free takes the next batch
    bool closing = false, reader_done = false;
    // Batches a GPU works on at once (guarded by mu).  `inflight` worker threads per GPU exist; `lane_limit` of them may hold a
    // batch.  With -inflight given the two are the same.  Without it four workers are there and two may work -- what a run
    // bound by its writer wants: more batches in flight only add memory traffic (25 M x 150 bp pairs: 1.67 / 1.78 / 1.85 s at
    // 2 / 3 / 4) -- until the writer turns out to be waiting for the GPU (a data set whose batches are bound by their slowest
    // reads: 5 % errors, k = 31, 25 M reads: loop 3.7 / 2.7 / 2.6 s at 2 / 3 / 4): then the limit goes up to `inflight`, after two
    // such batches in a row.  Each batch in flight runs in its own slot lane of the GPU's context (rcorrector_amd.h: rc_submit).
    int lane_limit = 2;
    int batches_done = 0;
    bool adaptive = false;
    std::vector<int> active;  // per GPU
    uint64_t total_reads = 0, total_cor = 0;  // UpdateSummary, main.cpp:73-79
};

// keep = false: the counting pass of a run in two passes (.gz inputs, inputs beyond the memory test, several GPUs): the same
// reader -- both mates' files side by side, parallel block reads, page-locked staging -- over sources of its own; the blocks
// are recycled instead of kept, and the counter releases the arenas when it has counted them.
void ingest_resident(Run &R, size_t batch_reads, int64_t *stored, bool keep);
// The same in two steps: the reader -- files into text blocks of host memory, `depth` of them ahead of the consumer -- needs no
// GPU, and `rcorrector` starts it before the contexts exist (HIP takes 0.08-0.25 s to come up, a tenth of a run on 25 M
// reads); consume() is the rest (index, upload, count, table).  abort(): the run will not be a one-pass run after all.
struct Ingest {
    Run &R;
    size_t batch_reads;
    bool keep;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Retained>> q;
    std::vector<std::unique_ptr<Retained>> spare;  // keep = false: blocks to fill again
    bool done = false;
    std::atomic<bool> stop{false};
    size_t depth = 3;
    std::thread reader;
    Ingest(Run &run, size_t batch, bool keep_) : R(run), batch_reads(batch), keep(keep_) {}
    void start();
    void consume(int64_t *stored);
    void abort();
};

// the batch buffers of the pipeline -- text blocks, page-locked arenas, output slices -- allocated, sized from the head of
// the first input, touched and registered with the GPU runtime (runs on a thread of its own while the table loads)
struct HeadStats {
    size_t nl = 0, last = 0, seq_len = 0;
};
HeadStats head_stats(const Run &R);
void warm_buffers(Run &R, const HeadStats &H);

// reader (the calling thread) -> workers -> writer; returns when the last batch is written and the workers have ended
void run_pipeline(Run &R);
