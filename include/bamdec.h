/* bamdec.h -- C ABI of libbamdec.so (alntools_amd/csrc/bamdec.c): the host side's BAM decoder.
 *
 * Stands where the reference iterates a pysam.AlignmentFile in Python (alntools/bam_utils.py:253-320): it hands the tuple encoder
 * column arrays instead of record objects.  Host only (plain C + zlib + pthreads); the device library libecb.so never touches
 * files.  Bound with ctypes in alntools_amd/bamdec.py.  Every function returns 0 or a negative BD_ERR_* code.
 */
#ifndef ALNTOOLS_AMD_BAMDEC_H
#define ALNTOOLS_AMD_BAMDEC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BD_OK          0
#define BD_ERR_IO     -1   /* cannot open / read the file */
#define BD_ERR_FORMAT -2   /* not BGZF / not BAM / truncated / CRC mismatch */
#define BD_ERR_MEM    -3
#define BD_ERR_ARG    -4

typedef struct bd_handle bd_handle;

int bd_abi_version(void);
/* Opens the file and reads the header (pysam.AlignmentFile(path), bam_utils.py:561); n_threads inflate BGZF blocks. */
int bd_open(const char* path, int n_threads, bd_handle** out);
void bd_close(bd_handle* h);
const char* bd_last_error(const bd_handle* h);
/* .references / .lengths / the header text (bam_utils.py:582, 615) */
int32_t bd_n_references(const bd_handle* h);
const char* bd_reference_name(const bd_handle* h, int32_t i);
int32_t bd_reference_length(const bd_handle* h, int32_t i);
const char* bd_header_text(const bd_handle* h);
/* All references in one call (a transcriptome BAM has hundreds of thousands): the names back to back, each with its NUL
 * (*blob_len bytes in all), and the lengths; both owned by the handle. */
int bd_references(bd_handle* h, const char** blob, size_t* blob_len, const int32_t** lens);
/* Up to max_records records in file order into caller-owned arrays of max_records elements; *n_out = 0 at the end of the file.
 * flag, tid (refID), pos, next_tid, next_pos: the raw BAM fields (alignment.flag, reference_id, reference_start,
 * next_reference_id, next_reference_start).  valid[i] = 1 if the record passes the reference's filter (bam_utils.py:264-270).
 * head[i] = 1 if it is valid and its query name -- cut at its first space when trim != 0 and that space is not the first
 * character (bam_utils.py:292-294) -- differs from the previous valid record's: the first alignment of a read
 * (bam_utils.py:289-320).  The previous name is kept across calls. */
int bd_read(bd_handle* h, size_t max_records, int trim, uint16_t* flag, int32_t* tid, int32_t* pos, int32_t* next_tid,
            int32_t* next_pos, uint8_t* valid, uint8_t* head, size_t* n_out);
/* The same records as the tuples ecb_push takes (include/ecb.h), without a pass through host arrays of BAM fields: read_id
 * counts the read heads (*cur: the id of the latest read started, 0xFFFFFFFF before the first; carried across calls by the
 * caller), locus / haplotype come from tid2locus / tid2hap (n_ref entries: the header maps of bam_utils.py:561-633) for valid
 * records and from reference 0 for the others, hapflag = flag & 0xFFF | ECB_FLAG_MATE_OTHER_REF | ECB_FLAG_NEXT_POS_NEG |
 * haplotype << 16, pos = reference_start.  *n_valid = valid records among the *n_out written. */
int bd_read_tuples(bd_handle* h, size_t max_records, int trim, const uint32_t* tid2locus, const uint32_t* tid2hap, int32_t n_ref,
                   uint32_t* cur, uint32_t* read_id, uint32_t* locus, uint32_t* hapflag, int32_t* pos, size_t* n_out, size_t* n_valid);

/* The multisample path's scan (alntools/bam_utils_multisample.py:257-300): as bd_read, but newrun[i] follows that path's run rule
 * -- the tracked name starts cut at its first space and becomes the WHOLE name of every record that starts a later run (:288-292)
 * -- and the cell barcode of every run started (field 14 of the tracked name split at "|||", :270-280) is kept for bd_ms_cells:
 * n cells, cell k = bytes [off[k], off[k + 1]), valid until the next call.  Do not mix with bd_read on one handle. */
int bd_read_ms(bd_handle* h, size_t max_records, uint16_t* flag, int32_t* tid, int32_t* pos, int32_t* next_tid, int32_t* next_pos,
               uint8_t* valid, uint8_t* newrun, size_t* n_out);
int bd_ms_cells(const bd_handle* h, const uint8_t** bytes, const uint32_t** off, size_t* n);
/* How far the records handed out so far reach into the file (compressed bytes; what has been read ahead and inflated but not
 * parsed yet is not counted) and the file's size -- a monotone measure of progress, for a reader that deals a file's records out
 * to several consumers in order (ABI 4; ABI 3 counted the read-ahead). */
int bd_progress(bd_handle* h, uint64_t* consumed, uint64_t* total);

#ifdef __cplusplus
}
#endif
#endif
