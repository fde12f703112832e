// Line-oriented reading and writing of (possibly gzip-compressed) text files for the genome-coordinate tools
// (mapLinesToGenome, combineCounts).  The reference reads and writes these files through gzstream's igzstream /
// ogzstream (reference lib/gzstream/gzstream.h, used at src/tools/GenomeGetter.hpp:39-40 and
// src/tools/combineCounts.cpp:100-101,184-185), i.e. zlib's gzopen / gzread / gzwrite: plain text is read
// transparently, output is a gzip stream at zlib's default level.  This header talks to zlib directly.
#ifndef HAMMLET_GZ_LINES_HPP
#define HAMMLET_GZ_LINES_HPP

#include <zlib.h>

#include <cstring>
#include <string>

namespace hammlet {

// std::getline over a gzFile: a line is everything up to '\n' (not stored); the last line needs no '\n'.  The string
// is treated as an istream treats it: the call that finds the file exhausted leaves it empty, and every call on a
// stream that has already failed (never opened, or exhausted before) returns false WITHOUT touching it - the reference's
// combineCounts goes on calling atoi on whatever the string then holds.
class GzLines {
    gzFile f_ = nullptr;
    char buf_[1 << 16];
    int have_ = 0, at_ = 0;
    bool failed_ = true;
    bool fill() {
        have_ = gzread(f_, buf_, (unsigned)sizeof(buf_));
        at_ = 0;
        return have_ > 0;
    }

public:
    GzLines() = default;
    GzLines(const GzLines&) = delete;
    GzLines& operator=(const GzLines&) = delete;
    ~GzLines() { close(); }
    bool open(const std::string& path) {
        close();
        f_ = gzopen(path.c_str(), "rb");
        have_ = at_ = 0;
        failed_ = (f_ == nullptr);
        return f_ != nullptr;
    }
    void close() {
        if (f_) gzclose(f_);
        f_ = nullptr;
        failed_ = true;
    }
    bool next(std::string& line) {
        if (failed_) return false;
        line.clear();
        bool any = false;
        for (;;) {
            if (at_ >= have_ && !fill()) {
                failed_ = true;   // (also when the file ends inside its last line: that line is delivered, the next call fails)
                return any;
            }
            any = true;
            const char* p = buf_ + at_;
            const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(have_ - at_)));
            if (nl) {
                line.append(p, (size_t)(nl - p));
                at_ += (int)(nl - p) + 1;
                return true;
            }
            line.append(p, (size_t)(have_ - at_));
            at_ = have_;
        }
    }
};

// ogzstream's counterpart: text appended to a gzip file
class GzOut {
    gzFile f_ = nullptr;
    std::string pending_;
    void flush() {
        if (f_ && !pending_.empty()) gzwrite(f_, pending_.data(), (unsigned)pending_.size());
        pending_.clear();
    }

public:
    GzOut() = default;
    GzOut(const GzOut&) = delete;
    GzOut& operator=(const GzOut&) = delete;
    ~GzOut() { close(); }
    bool open(const std::string& path) {
        close();
        f_ = gzopen(path.c_str(), "wb");
        return f_ != nullptr;
    }
    void write(const std::string& text) {
        pending_ += text;
        if (pending_.size() >= (1u << 16)) flush();
    }
    void close() {
        flush();
        if (f_) gzclose(f_);
        f_ = nullptr;
    }
};

}  // namespace hammlet
#endif
