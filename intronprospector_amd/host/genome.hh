// genome.hh -- indexed FASTA -> device-ready 2-bit genome.
//
// Stands for the reference's Genome (src/genome.hh:15-25, src/genome.cc:44-74):
// the constructor needs "<fasta>.fai" (fai_load3 without FAI_CREATE), has_seq()
// answers by name.  Instead of fetching 2+2 bases per new junction from the
// file, the whole genome is packed once into 2-bit codes plus a non-ACGT mask
// (the layout ipx_set_genome_2bit takes) and the motif test runs on the GPU;
// raw() returns original bytes for the rare motif that holds a non-ACGT base.
#ifndef IPX_HOST_GENOME_HH
#define IPX_HOST_GENOME_HH
#include <cstdint>
#include <string>
#include <vector>

namespace ipxhost {

class Genome {
public:
    explicit Genome(const std::string &genome_fa);    // throws std::runtime_error like genome.cc:47-49
    ~Genome();
    Genome(const Genome &) = delete;
    Genome &operator=(const Genome &) = delete;

    bool has_seq(const std::string &chrom) const;
    size_t n_seqs() const { return seqs_.size(); }
    const std::string &name(size_t i) const { return seqs_[i].name; }
    int64_t length(size_t i) const { return seqs_[i].len; }
    const uint8_t *twobit(size_t i) const { return seqs_[i].two.data(); }
    const uint8_t *mask(size_t i) const { return seqs_[i].mask.data(); }
    // byte at 0-based position pos of sequence chrom, as faidx would return it
    char raw(const std::string &chrom, int64_t pos) const;

private:
    struct Seq {
        std::string name;
        int64_t len = 0, offset = 0;
        int line_blen = 0, line_len = 0;
        std::vector<uint8_t> two, mask;
    };
    std::vector<Seq> seqs_;
    const uint8_t *data_ = nullptr;     // mmap of the FASTA file
    size_t size_ = 0;
    int fd_ = -1;
    const Seq *find(const std::string &chrom) const;
};

}  // namespace ipxhost
#endif
