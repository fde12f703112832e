// Records - the output files of a run (reference src/Records.hpp:20-243, src/StateMarginals.hpp:268-310):
//   PREFIXmarginalsSUFFIX    one line per marginal segment: SIZE \t count_0 ... count_{Kmax}   (written at close)
//   PREFIXsequencesSUFFIX    one line per recorded sweep: tab-separated SIZE:STATE segments
//   PREFIXblocksSUFFIX       one line per recorded sweep: tab-separated block sizes
//   PREFIXparametersSUFFIX   one line per recorded sweep: mean \t var per state
//   PREFIXcompressionSUFFIX  one line per recorded sweep: T / #blocks
//   PREFIXsegmentsSUFFIX     one line per recorded sweep: #marginal segments \t internal size (see note)
//   PREFIXmaxsegmentationSUFFIX  (extension, `-O X`) what the reference's maxSegmentation tool prints for the marginals
//                            file (reference src/tools/maxSegmentation.cpp:53-82), computed on the device at close
// The marginal counts are accumulated on the device (difference arrays + boundary bitmap) and fetched in
// run-length form at close(); the per-sweep files are appended from the device's block list and state
// sequence after every recorded sweep.
// Note on "segments": the reference's second column is the length of its internal count queue
// (StateMarginals::internalSize(), src/StateMarginals.hpp:204) at the moment Records::record writes the line
// (src/Records.hpp:208-209) - before the sweep's LAST run of equal states has been added.  The queue holds one
// record per marginal segment - the non-zero counts in state order, an index entry before every stored state
// whose predecessor is not stored (never before state 0), a terminator (src/StateMarginals.hpp:71-115) - so a
// record's length is a function of the SET of states with a count: |S| + #{s in S: s > 0, s-1 not in S} + 1.
// MarginalSegmentSets below keeps those sets (a sorted list of segment starts with a 64-bit state mask each,
// merged with the sweep's runs in one pass) and yields the reference's two numbers exactly.
#ifndef HAMMLET_RECORDS_HPP
#define HAMMLET_RECORDS_HPP

#include <fstream>
#include <string>
#include <vector>

namespace hammlet {

inline bool fileExists(const std::string& path) {
    std::ifstream f(path.c_str());
    return f.good();
}

// marginal segments as (start, set of states with a non-zero count): what the `segments` file needs and nothing more
class MarginalSegmentSets {
    std::vector<uint32_t> mStart{0};
    std::vector<uint64_t> mStates{0};
    static uint64_t recordLength(uint64_t set) {
        return (uint64_t)__builtin_popcountll(set) + (uint64_t)__builtin_popcountll(set & ~(set << 1) & ~1ull) + 1;
    }

public:
    size_t nrSegments() const { return mStart.size(); }
    // one recorded sweep given as runs of equal state (start of run r, its state; the last run ends at T).  Returns the
    // length of the reference's count queue when all runs but the last have been added.
    uint64_t addSweep(const std::vector<uint32_t>& runStart, const std::vector<int16_t>& runState, uint64_t T) {
        const size_t R = runStart.size(), M = mStart.size();
        std::vector<uint32_t> start;
        std::vector<uint64_t> states;
        start.reserve(M + R);
        states.reserve(M + R);
        uint64_t queued = 0;
        size_t r = 0, m = 0;
        for (uint64_t pos = 0; pos < T;) {
            const uint64_t nextRun = r + 1 < R ? runStart[r + 1] : T, nextSeg = m + 1 < M ? mStart[m + 1] : T;
            const uint64_t with = mStates[m] | (1ull << runState[r]);
            start.push_back((uint32_t)pos);
            states.push_back(with);
            queued += recordLength(pos < runStart[R - 1] ? with : mStates[m]);
            pos = nextRun < nextSeg ? nextRun : nextSeg;
            r += pos == nextRun;
            m += pos == nextSeg;
        }
        mStart.swap(start);
        mStates.swap(states);
        return queued;
    }
};

class Records {
    const size_t mSize;
    std::string mPrefix, mSuffix;
    hml_ctx* mCtx = nullptr;
    bool mRecordMarginals = true, mRecordBlocks = false, mRecordCompression = false, mRecordSequences = false,
         mRecordTheta = false, mRecordSegments = false, mRecordMaxSeg = false, mAccumulate = false;
    std::ofstream mMarginalsFile, mSequenceFile, mBlocksFile, mThetaFile, mCompressionsFile, mSegmentFile, mMaxSegFile;
    bool mClosed = false;
    MarginalSegmentSets mSegmentSets;   // only maintained when the segments file is requested

    void setRecordX(std::ofstream& file, const std::string& type, bool& member, bool flag, bool overwrite) {
        member = flag;
        if (member && !file.is_open()) {
            const std::string filename = mPrefix + type + mSuffix;
            if (fileExists(filename) && !overwrite)
                throw std::runtime_error("File " + filename + " already exists! Use -w to allow overwrite!");
            file.open(filename.c_str());
            if (!file.is_open()) throw std::runtime_error("Cannot write to file " + filename + "!");
        }
    }

public:
    Records(const Records&) = delete;
    Records(size_t T, std::string prefix, std::string suffix, const size_t /*nrStates*/)
        : mSize(T), mPrefix(std::move(prefix)), mSuffix(std::move(suffix)) {}
    ~Records() {
        try { close(); } catch (...) {}
    }
    void attach(hml_ctx* ctx) { mCtx = ctx; }   // sampleHMM / StateSequence::sample do this; the marginals are fetched from it at close()

    void setRecordMarginals(bool b, bool overwrite = false) { setRecordX(mMarginalsFile, "marginals", mRecordMarginals, b, overwrite); }
    void setRecordBlocks(bool b, bool overwrite = false) { setRecordX(mBlocksFile, "blocks", mRecordBlocks, b, overwrite); }
    void setRecordCompression(bool b, bool overwrite = false) { setRecordX(mCompressionsFile, "compression", mRecordCompression, b, overwrite); }
    void setRecordStateSequence(bool b, bool overwrite = false) { setRecordX(mSequenceFile, "sequences", mRecordSequences, b, overwrite); }
    void setRecordTheta(bool b, bool overwrite = false) { setRecordX(mThetaFile, "parameters", mRecordTheta, b, overwrite); }
    void setRecordSegments(bool b, bool overwrite = false) { setRecordX(mSegmentFile, "segments", mRecordSegments, b, overwrite); }

    void setRecordMaxSegmentation(bool b, bool overwrite = false) { setRecordX(mMaxSegFile, "maxsegmentation", mRecordMaxSeg, b, overwrite); }

    // chains whose marginals only feed a pool: accumulated on the device, written by another chain's Records
    void setAccumulateMarginals(bool b) { mAccumulate = b; }
    // give up the marginals files (a pooling step failed): they are closed empty
    void discardMarginals() { mCtx = nullptr; mRecordMarginals = false; mRecordMaxSeg = false; }
    bool recordsMarginals() const { return mRecordMarginals || mRecordMaxSeg || mAccumulate; }
    bool needsPerSweepData() const { return mRecordBlocks || mRecordCompression || mRecordSequences || mRecordTheta || mRecordSegments; }

    // one recorded sweep: Records::record(state, N) for every block in order (reference src/Records.hpp:155-235) ...
    void recordStates(hml_ctx* ctx) {
        if (!(mRecordBlocks || mRecordCompression || mRecordSequences || mRecordSegments)) return;
        uint64_t B = 0;
        hml_check(hml_get_num_blocks(ctx, &B));
        std::vector<uint32_t> starts(B + 1);
        std::vector<int16_t> q(B);
        hml_check(hml_get_blocks(ctx, starts.data()));
        hml_check(hml_get_states(ctx, q.data()));
        bool firstSeg = true;
        size_t segStart = 0;
        std::vector<uint32_t> runStart;
        std::vector<int16_t> runState;
        for (size_t b = 0; b < B; ++b) {
            if (mRecordBlocks) mBlocksFile << (b ? "\t" : "") << (starts[b + 1] - starts[b]);
            const bool last = (b + 1 == B);
            if (last || q[b + 1] != q[b]) {
                if (mRecordSequences) mSequenceFile << (firstSeg ? "" : "\t") << (starts[b + 1] - starts[segStart]) << ":" << (size_t)q[b];
                if (mRecordSegments) { runStart.push_back(starts[segStart]); runState.push_back(q[b]); }
                firstSeg = false;
                segStart = b + 1;
            }
        }
        if (mRecordBlocks) mBlocksFile << "\n";
        if (mRecordSequences) mSequenceFile << "\n";
        if (mRecordCompression) mCompressionsFile << ((double)mSize) / ((double)B) << std::endl;
        if (mRecordSegments) {
            // (the reference adds to its marginals only when they are recorded, src/Records.hpp:176,212: otherwise one
            // empty record)
            if (recordsMarginals()) {
                const uint64_t queued = mSegmentSets.addSweep(runStart, runState, mSize);
                mSegmentFile << mSegmentSets.nrSegments() << "\t" << queued << std::endl;
            } else {
                mSegmentFile << 1 << "\t" << 1 << std::endl;
            }
        }
    }
    // ... and Records::record(theta) (src/Records.hpp:196-203): the parameters line of the sweep
    template <typename ThetaT>
    void record(const ThetaT& theta) {
        if (mRecordTheta) mThetaFile << theta.str() << std::endl;
    }
    template <typename ThetaT>
    void recordSweep(hml_ctx* ctx, const ThetaT& theta) {
        recordStates(ctx);
        record(theta);
    }

    void close() {
        if (mClosed) return;
        mClosed = true;
        if (mRecordMarginals && mMarginalsFile.is_open()) {
            if (mCtx) {
                uint64_t n = 0;
                int cols = 0;
                hml_check(hml_marginals_rle(mCtx, &n, &cols, nullptr, nullptr));
                std::vector<uint64_t> seg(n);
                std::vector<int32_t> cnt((size_t)n * (cols > 0 ? cols : 1));
                hml_check(hml_marginals_rle(mCtx, &n, &cols, seg.data(), cols > 0 ? cnt.data() : nullptr));
                std::string out;
                out.reserve((size_t)n * (8 + 4 * (cols > 0 ? cols : 0)));
                for (uint64_t i = 0; i < n; ++i) {
                    out += std::to_string(seg[i]);
                    for (int s = 0; s < cols; ++s) { out += '\t'; out += std::to_string(cnt[i * cols + s]); }
                    out += '\n';
                }
                mMarginalsFile << out;
            } else {
                mMarginalsFile << mSize << "\n";
            }
            mMarginalsFile.close();
        }
        if (mRecordMaxSeg && mMaxSegFile.is_open()) {
            // the tool's output format: the running state starts at 0, so a first run of another state is preceded by "0\t0"
            if (mCtx) {
                uint64_t n = 0;
                hml_check(hml_max_segmentation(mCtx, &n, nullptr, nullptr));
                std::vector<uint64_t> len(n);
                std::vector<int32_t> st(n);
                hml_check(hml_max_segmentation(mCtx, &n, len.data(), st.data()));
                if (n && st[0] != 0) mMaxSegFile << 0 << "\t" << 0 << "\n";
                for (uint64_t i = 0; i < n; ++i) mMaxSegFile << len[i] << "\t" << st[i] << "\n";
            } else {
                mMaxSegFile << mSize << "\t" << 0 << "\n";
            }
            mMaxSegFile.close();
        }
        if (mSequenceFile.is_open()) mSequenceFile.close();
        if (mBlocksFile.is_open()) mBlocksFile.close();
        if (mThetaFile.is_open()) mThetaFile.close();
        if (mCompressionsFile.is_open()) mCompressionsFile.close();
        if (mSegmentFile.is_open()) mSegmentFile.close();
    }
};

}  // namespace hammlet
#endif
