// colord_api.cpp — colord::DecompressionStream (include/colord_api.h; the reference's src/API/colord_api.cpp:166-420) on the
// record stream of cli/reader.hpp.  Built into colord_amd/libcolord_hip_api.a; link with -lcolord_hip -lpthread.
#include "../../../include/colord_api.h"
#include "../cli/reader.hpp"
#include <ctime>
#include <sstream>

namespace colord
{
	static const char* source_name(ReadsSource s) { return s == ReadsSource::ONT ? "Oxford Nanopore" : s == ReadsSource::PBRaw ? "PacBio raw" : "PacBio HiFi"; }
	static const char* quality_name(QualityCompressionMode m)
	{
		static const char* n[] = { "Original", "Quinary average", "Quad average", "Binary average", "Quinary threshold", "Quad threshold", "Binary threshold", "Average", "None" };
		return n[(int)m];
	}
	static const char* header_name(HeaderCompressionMode m) { return m == HeaderCompressionMode::Original ? "Original" : m == HeaderCompressionMode::Main ? "Main" : "None"; }

	// the lines of the reference's Info::ToOstream (colord_api.cpp:83-101), spelling included
	void Info::ToOstream(std::ostream& oss) const
	{
		time_t t = (time_t)time;
		oss << "is fastq: " << std::boolalpha << isFastq << "\n"
		    << "colord archive version: " << versionMajor << "." << versionMinor << "." << versionPatch << "\n"
		    << "total reads: " << totalReads << "\n"
		    << "colord archive creaton datetime: " << asctime(localtime(&t))
		    << "command line used to create colord archive: " << fullCommandLine << "\n"
		    << "compression level: " << compressionLevel << "\n"
		    << "reads source: " << source_name(readsSource) << "\n"
		    << "quality compression mode: " << quality_name(qualityCompressionMode) << "\n"
		    << "header compression mode: " << header_name(headerCompressionMode) << "\n"
		    << "quality reverse thresholds: ";
		for (auto v : qualityReverseThresholds) oss << v << " ";
		oss << "\n";
	}

	class DecompressionStream::DecompressionStreamImpl
	{
	public:
		colord_hip_reader::RecordStream rs;
		Info info;
		explicit DecompressionStreamImpl(const std::string& path, const std::string& genome = "") : rs(path, genome)
		{
			const auto& I = rs.info(); const auto& M = rs.meta();
			info.isFastq = rs.is_fastq();
			info.versionMajor = I.version_major; info.versionMinor = I.version_minor; info.versionPatch = I.version_patch;
			info.totalBytes = I.total_bytes; info.totalBases = I.total_bases; info.totalReads = I.total_reads; info.time = I.time;
			info.fullCommandLine = I.command_line;
			info.compressionLevel = M.level;
			// the `meta` stream's codes (params.h: DataSource ONT 0, PBRaw 1, PBHiFi 2; QualityComprMode and HeaderComprMode in enum order)
			if (M.source > 2 || M.qual_mode > 8 || M.header_mode > 2) throw std::runtime_error("unknown mode code in the archive's `meta` stream");
			info.readsSource = M.source == 0 ? ReadsSource::ONT : M.source == 1 ? ReadsSource::PBRaw : ReadsSource::PBHiFi;
			info.qualityCompressionMode = (QualityCompressionMode)M.qual_mode;
			info.headerCompressionMode = (HeaderCompressionMode)M.header_mode;
			info.qualityReverseThresholds = M.rev;
		}
	};

	DecompressionStream::DecompressionStream(const std::string& inputFilePath) : pImpl(new DecompressionStreamImpl(inputFilePath)) {}
	DecompressionStream::DecompressionStream(const std::string& inputFilePath, const std::string& refGenomePath) : pImpl(new DecompressionStreamImpl(inputFilePath, refGenomePath)) {}
	DecompressionStream::~DecompressionStream() = default;
	Info DecompressionStream::GetInfo() const { return pImpl->info; }
	DecompressionRecord DecompressionStream::NextRecord()
	{
		DecompressionRecord res;
		colord_hip_reader::Record r;
		if (!pImpl->rs.next(r)) { res.at_end = true; return res; }
		res.header_.assign((const char*)r.header, r.header_len);
		res.read_.resize(r.n_bases);
		for (size_t i = 0; i < r.n_bases; ++i) res.read_[i] = "ACGTN"[(r.bases[i] & 7) > 4 ? 4 : (r.bases[i] & 7)];
		if (r.quals)
		{
			if (r.plus_is_header) res.qual_header_ = res.header_;
			res.qual_.assign((const char*)r.quals, r.n_bases);
		}
		return res;
	}
}
