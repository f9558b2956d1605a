/* colord_api.h — public C++ reader API of the MI355X-native CoLoRd build: the surface of the reference's
 * src/API/colord_api.h:27-103 (namespace colord: DecompressionRecord, Info and its enums, DecompressionStream with GetInfo()
 * and NextRecord()), so that a program written against the reference's libcolord_api.a — e.g. its src/API_example/
 * api_example.cpp — compiles and runs unchanged against this one:
 *     g++ -std=c++17 prog.cpp -I include -L colord_amd -lcolord_hip_api -lcolord_hip -lz -lpthread
 * Archives of the reference and of colord_hip (incl. multi-GPU archives with a `hipdomains` stream) are read alike; the
 * decoders are the host functions of the C ABI (include/colord_hip.h, a17), no GPU is needed.  Errors are std::runtime_error,
 * as in the reference.  Archives written with -G but without -s take the genome's path in the second constructor. */
#pragma once
#include <cstdint>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace colord
{
	/* one FASTQ / FASTA record; converts to false after the last one */
	class DecompressionRecord
	{
		friend class DecompressionStream;
		bool at_end = false;
		std::string header_, read_, qual_header_, qual_;
	public:
		operator bool() const { return !at_end; }
		const std::string& ReadHeader() const { return header_; }
		const std::string& Read() const { return read_; }
		const std::string& QualHeader() const { return qual_header_; }      /* the header again when the input repeated it after '+', else empty */
		const std::string& Qual() const { return qual_; }
	};

	enum class ReadsSource { ONT, PBRaw, PBHiFi };
	enum class QualityCompressionMode { Original, QuinaryAverage, QuadAverage, BinaryAverage, QuinaryThreshold, QuadThreshold, BinaryThreshold, Average, None };
	enum class HeaderCompressionMode { Original, Main, None };

	/* what `colord info` knows plus the coding parameters of the `meta` stream */
	struct Info
	{
		bool isFastq;
		uint32_t versionMajor, versionMinor, versionPatch;
		uint64_t totalBytes, totalBases;
		uint32_t totalReads;
		uint64_t time;
		std::string fullCommandLine;
		int32_t compressionLevel{};
		ReadsSource readsSource{};
		QualityCompressionMode qualityCompressionMode{};
		HeaderCompressionMode headerCompressionMode{};
		std::vector<uint32_t> qualityReverseThresholds;
		void ToOstream(std::ostream& oss) const;
	};

	class DecompressionStream
	{
		class DecompressionStreamImpl;
		const std::unique_ptr<DecompressionStreamImpl> pImpl;
	public:
		explicit DecompressionStream(const std::string& inputFilePath);
		explicit DecompressionStream(const std::string& inputFilePath, const std::string& refGenomePath);
		Info GetInfo() const;
		DecompressionRecord NextRecord();
		~DecompressionStream();
	};
}
