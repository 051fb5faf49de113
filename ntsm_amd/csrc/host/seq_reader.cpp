#include "seq_reader.hpp"

#include <cctype>
#include <cstdlib>
#include <cstring>

namespace ntsm {

bool SeqReader::open(const std::string &path, uint64_t offset)
{
	close();
	if (offset == 0 && !getenv("NTSM_ZLIB_ONLY") && GzStream::is_gzip(path)) {
		gz_.reset(new GzStream());
		if (!gz_->open(path)) { gz_.reset(); return false; }
	} else {
		f_ = gzopen(path.c_str(), "r");
		if (!f_) return false;
		gzbuffer(f_, 1 << 20);
		if (offset && gzseek(f_, (z_off_t) offset, SEEK_SET) < 0) { close(); return false; }
	}
	buf_.resize(kBuf);
	beg_ = end_ = 0;
	eof_ = false;
	pending_ = 0;
	return true;
}

bool SeqReader::open_stream(std::unique_ptr<GzStream> gz)
{
	close();
	if (!gz) return false;
	gz_ = std::move(gz);
	buf_.resize(kBuf);
	beg_ = end_ = 0;
	eof_ = false;
	pending_ = 0;
	return true;
}

int SeqReader::source_read(void *dst, unsigned len)
{
	return gz_ ? gz_->read(dst, len) : gzread(f_, dst, len);
}

void SeqReader::close()
{
	if (f_) gzclose(f_);
	f_ = nullptr;
	gz_.reset();
}

bool SeqReader::refill()
{
	beg_ = 0;
	end_ = source_read(buf_.data(), kBuf);
	if (end_ == 0) { eof_ = true; return false; }
	if (end_ < 0) { eof_ = true; return false; }         /* end_ stays negative: sticky error */
	return true;
}

int SeqReader::get()
{
	if (end_ < 0) return -3;
	if (beg_ >= end_) {
		if (eof_) return -1;
		if (!refill()) return end_ < 0 ? -3 : -1;
	}
	return buf_[beg_++];
}

int64_t SeqReader::until(bool line, std::vector<char> &dst, int *delim)
{
	bool touched = false;
	if (delim) *delim = 0;
	for (;;) {
		if (end_ < 0) return -3;
		if (beg_ >= end_) {
			if (eof_) break;
			if (!refill()) {
				if (end_ < 0) return -3;
				break;
			}
		}
		int i = beg_;
		if (line) {
			const void *nl = memchr(buf_.data() + beg_, '\n', (size_t) (end_ - beg_));
			i = nl ? (int) ((const unsigned char *) nl - buf_.data()) : end_;
		} else {
			while (i < end_ && !isspace(buf_[i])) ++i;
		}
		touched = true;
		dst.insert(dst.end(), buf_.data() + beg_, buf_.data() + i);
		beg_ = i + 1;
		if (i < end_) {
			if (delim) *delim = buf_[i];
			break;
		}
	}
	if (!touched && eof_ && beg_ >= end_) return -1;
	if (line && dst.size() > 1 && dst.back() == '\r') dst.pop_back();
	return (int64_t) dst.size();
}

/* One whole record with plain '\n' line ends inside the buffer:
 *   FASTQ: header \n SEQ \n +... \n QUAL \n   with |QUAL| == |SEQ| >= 1
 *   FASTA: header \n SEQ \n  followed by a line that starts a new record ('>' or '@')
 * Anything else (CR, wrapped lines, empty sequence, record cut by the buffer end, end of file)
 * is left to the general path, which reproduces kseq byte for byte. */
bool SeqReader::fast_record(int64_t *len)
{
	for (int attempt = 0; attempt < 2; ++attempt) {
		const unsigned char *b = buf_.data();
		const unsigned char *p = b + beg_, *e = b + end_;
		const unsigned char *nl1 = p < e ? (const unsigned char *) memchr(p, '\n', (size_t) (e - p)) : nullptr;
		const unsigned char *q = nl1 ? nl1 + 1 : nullptr;
		const unsigned char *nl2 = (q && q < e) ? (const unsigned char *) memchr(q, '\n', (size_t) (e - q)) : nullptr;
		if (nl2 && nl2 + 1 < e) {
			const int64_t slen = nl2 - q;
			if (slen < 1 || q[0] == '>' || q[0] == '@' || q[0] == '+' || nl2[-1] == '\r' || nl1[-1 + (nl1 == p)] == '\r') return false;
			const unsigned char c3 = nl2[1];
			const unsigned char *name_end = p;
			while (name_end < nl1 && !isspace(*name_end)) ++name_end;
			if (c3 == '>' || c3 == '@') {                       /* single-line FASTA record */
				name_.assign((const char *) p, (size_t) (name_end - p));
				seq_ptr_ = (const char *) q;
				*len = slen;
				pending_ = c3;
				beg_ = (int) (nl2 + 2 - b);
				return true;
			}
			if (c3 != '+') return false;
			const unsigned char *nl3 = (const unsigned char *) memchr(nl2 + 1, '\n', (size_t) (e - (nl2 + 1)));
			const unsigned char *u = nl3 ? nl3 + 1 : nullptr;
			const unsigned char *nl4 = (u && u < e) ? (const unsigned char *) memchr(u, '\n', (size_t) (e - u)) : nullptr;
			if (nl4) {
				if (nl4 - u != slen || nl4[-1] == '\r') return false;
				name_.assign((const char *) p, (size_t) (name_end - p));
				seq_ptr_ = (const char *) q;
				*len = slen;
				pending_ = 0;
				beg_ = (int) (nl4 + 1 - b);
				return true;
			}
		}
		/* record not complete in the buffer: slide the tail to the front and read more, once */
		if (attempt == 1 || eof_ || end_ < 0) return false;
		const int keep = end_ - beg_;
		if (keep > kBuf / 2) return false;
		memmove(buf_.data(), buf_.data() + beg_, (size_t) keep);
		const int got = source_read(buf_.data() + keep, (unsigned) (kBuf - keep));
		beg_ = 0;
		if (got < 0) { end_ = keep; return false; }              /* the general path will hit the error again */
		if (got == 0) eof_ = true;
		end_ = keep + got;
	}
	return false;
}

int64_t SeqReader::next()
{
	int c;
	if (pending_ == 0) {
		while ((c = get()) >= 0 && c != '>' && c != '@') { }
		if (c < 0) return c;
		pending_ = c;
	}
	{
		int64_t flen;
		if (fast_record(&flen)) return flen;
	}
	seq_.clear();
	qual_len_ = 0;
	scratch_.clear();
	int64_t r = until(false, scratch_, &c);
	name_.assign(scratch_.data(), scratch_.size());
	if (r < 0) return r;
	if (c != '\n') {
		scratch_.clear();
		until(true, scratch_, nullptr);                  /* comment: ignored */
	}
	while ((c = get()) >= 0 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		seq_.push_back((char) c);
		until(true, seq_, nullptr);
	}
	if (c == '>' || c == '@') pending_ = c;
	seq_ptr_ = seq_.data();
	if (c != '+') return (int64_t) seq_.size();
	while ((c = get()) >= 0 && c != '\n') { }
	if (c == -1) return -2;
	qual_.clear();
	while (until(true, qual_, nullptr) >= 0 && qual_.size() < seq_.size()) { }
	pending_ = 0;
	seq_ptr_ = seq_.data();
	if (qual_.size() != seq_.size()) return -2;
	return (int64_t) seq_.size();
}

} // namespace ntsm
