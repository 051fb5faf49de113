#include "seq_reader.hpp"

#include <cctype>
#include <cstring>

namespace ntsm {

bool SeqReader::open(const std::string &path)
{
	close();
	f_ = gzopen(path.c_str(), "r");
	if (!f_) return false;
	gzbuffer(f_, 1 << 20);
	buf_.resize(kBuf);
	beg_ = end_ = 0;
	eof_ = false;
	pending_ = 0;
	return true;
}

void SeqReader::close()
{
	if (f_) gzclose(f_);
	f_ = nullptr;
}

bool SeqReader::refill()
{
	beg_ = 0;
	end_ = gzread(f_, buf_.data(), kBuf);
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

int64_t SeqReader::next()
{
	int c;
	if (pending_ == 0) {
		while ((c = get()) >= 0 && c != '>' && c != '@') { }
		if (c < 0) return c;
		pending_ = c;
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
	if (c != '+') return (int64_t) seq_.size();
	while ((c = get()) >= 0 && c != '\n') { }
	if (c == -1) return -2;
	qual_.clear();
	while (until(true, qual_, nullptr) >= 0 && qual_.size() < seq_.size()) { }
	pending_ = 0;
	if (qual_.size() != seq_.size()) return -2;
	return (int64_t) seq_.size();
}

} // namespace ntsm
