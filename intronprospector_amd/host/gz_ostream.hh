// gz_ostream.hh -- output stream that gzips when the file name ends in ".gz"
// (the behaviour of the reference's AutoGzipOutput, src/AutoGzip.hh:9-11,40-65).
#ifndef IPX_HOST_GZ_OSTREAM_HH
#define IPX_HOST_GZ_OSTREAM_HH
#include <fstream>
#include <memory>
#include <ostream>
#include <streambuf>
#include <string>
#include <zlib.h>

namespace ipxhost {

class GzStreamBuf : public std::streambuf {
public:
    explicit GzStreamBuf(const std::string &path) : gz_(gzopen(path.c_str(), "wb")) { setp(buf_, buf_ + sizeof buf_); }
    ~GzStreamBuf() override {
        sync();
        if (gz_) gzclose(gz_);
    }
    bool is_open() const { return gz_ != nullptr; }

protected:
    int_type overflow(int_type c) override {
        if (flush_buf() != 0) return traits_type::eof();
        if (!traits_type::eq_int_type(c, traits_type::eof())) {
            *pptr() = traits_type::to_char_type(c);
            pbump(1);
        }
        return traits_type::not_eof(c);
    }
    int sync() override { return flush_buf(); }

private:
    gzFile gz_;
    char buf_[1 << 16];
    int flush_buf() {
        const int n = (int)(pptr() - pbase());
        if (n > 0 && gz_ && gzwrite(gz_, pbase(), (unsigned)n) != n) return -1;
        setp(buf_, buf_ + sizeof buf_);
        return 0;
    }
};

class AutoGzipOutput : public std::ostream {
public:
    explicit AutoGzipOutput(const std::string &filename) : std::ostream(nullptr) {
        const bool gz = filename.size() >= 3 && filename.compare(filename.size() - 3, 3, ".gz") == 0;
        if (gz) {
            gz_.reset(new GzStreamBuf(filename));
            if (!gz_->is_open()) throw std::ios_base::failure("Failed to open gzip file: " + filename);
            rdbuf(gz_.get());
        } else {
            file_.reset(new std::ofstream(filename));
            if (!file_->is_open()) throw std::ios_base::failure("Failed to open file: " + filename);
            rdbuf(file_->rdbuf());
        }
    }

private:
    std::unique_ptr<GzStreamBuf> gz_;
    std::unique_ptr<std::ofstream> file_;
};

}  // namespace ipxhost
#endif
