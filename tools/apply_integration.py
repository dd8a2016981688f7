#!/usr/bin/env python3
"""The INTEGRATION.md §2 patch as a program: turns the reference's main.cpp (mourisl/Rcorrector
v1.0.7) into a host that keeps its own command line, file I/O (Reads.h) and output code and hands
the hot path -- table load, ERROR_RATE pass, per-batch correction -- to librcorrector_amd.so.

    tools/apply_integration.py /root/reference/main.cpp patched_main.cpp

It edits by line number (pinned to v1.0.7; a few tokens are checked so that a different version is
refused instead of mangled) and carries none of the reference's text: only what is INSERTED lives
here.  tests/test_integration_patch.py applies it to a temporary copy, compiles the result with the
reference's other sources and links it against the library (oracle/_ref/rcorrector_patched, where
/root/reference exists); the GPU suite then runs that binary on the golden fixtures.
"""
import sys

INCLUDES = '''#include <vector>
#include "rcorrector_amd.h"
static const char *rcDumpPath = NULL ;   // the -c argument, for rc_table_load_jfdump
static rc_ctx *rcGpu = NULL ;
'''

FORCE_BATCH_PATH = '''	// the batch loop (main.cpp:439-523) is the one that talks to the library; 64 x 512 reads per batch
	if ( numOfThreads < 64 )
		numOfThreads = 64 ;
'''

LOAD_TABLE = '''	int64_t kmerCount = 0 ;   // replaces the Store::Put loop, main.cpp:294-308
	{
		rc_config cfg = { 0, kmerLength, MAX_FIX_PER_K } ;
		char rcErr[512] ;
		rcGpu = rc_create( &cfg, rcErr, sizeof( rcErr ) ) ;
		if ( !rcGpu )
		{
			fprintf( stderr, "%s\\n", rcErr ) ;
			exit( 1 ) ;
		}
		if ( rc_table_load_jfdump( rcGpu, rcDumpPath, &kmerCount ) )
		{
			fprintf( stderr, "%s\\n", rc_last_error( rcGpu ) ) ;
			exit( 1 ) ;
		}
	}
	fprintf( stderr, "Stored %d kmers\\n", (int)kmerCount ) ;
'''

ERROR_RATE = '''	// replaces the second scan of the dump, main.cpp:314-357
	if ( rc_estimate_error_rate( rcGpu, errorRateKmerPortion, &ERROR_RATE ) )
	{
		fprintf( stderr, "%s\\n", rc_last_error( rcGpu ) ) ;
		exit( 1 ) ;
	}
'''

RUN_PARAMS = '''	rc_set_run_params( rcGpu, ERROR_RATE, badQualityThreshold ) ;
'''

CORRECT_BATCH = '''			{	// replaces pthread_create / ErrorCorrection_Thread / pthread_join, main.cpp:479-483:
				// the batch of struct _Read (Reads.h:13-23) packed SoA, one rc_correct_batch, results back
				std::vector<char> rcSeq, rcQual, rcSeq2, rcQual2 ;
				std::vector<uint32_t> rcOff( 1, 0 ), rcOff2( 1, 0 ) ;
				const bool rcPaired = arg.readBatch2 != NULL ;
				for ( i = 0 ; i < batchSize ; ++i )
				{
					size_t len = strlen( readBatch[i].seq ) ;
					rcSeq.insert( rcSeq.end(), readBatch[i].seq, readBatch[i].seq + len + 1 ) ;
					rcQual.insert( rcQual.end(), readBatch[i].qual, readBatch[i].qual + len ) ;
					rcQual.push_back( 0 ) ;
					rcOff.push_back( (uint32_t)rcSeq.size() ) ;
					if ( rcPaired )
					{
						len = strlen( readBatch2[i].seq ) ;
						rcSeq2.insert( rcSeq2.end(), readBatch2[i].seq, readBatch2[i].seq + len + 1 ) ;
						rcQual2.insert( rcQual2.end(), readBatch2[i].qual, readBatch2[i].qual + len ) ;
						rcQual2.push_back( 0 ) ;
						rcOff2.push_back( (uint32_t)rcSeq2.size() ) ;
					}
				}
				std::vector<int32_t> rcRet( 2 * batchSize ), rcL( 2 * batchSize ), rcM( 2 * batchSize ), rcH( 2 * batchSize ) ;
				rc_batch rcB ;
				memset( &rcB, 0, sizeof( rcB ) ) ;
				rcB.mode = rcPaired ? 1 : ( arg.interleaved ? 2 : 0 ) ;
				rcB.n = (size_t)batchSize ;
				rcB.seq = rcSeq.data() ; rcB.qual = rcQual.data() ; rcB.off = rcOff.data() ;
				rcB.seq2 = rcSeq2.data() ; rcB.qual2 = rcQual2.data() ; rcB.off2 = rcOff2.data() ;
				rcB.ret = rcRet.data() ; rcB.l = rcL.data() ; rcB.m = rcM.data() ; rcB.h = rcH.data() ;
				if ( rc_correct_batch( rcGpu, &rcB ) )
				{
					fprintf( stderr, "%s\\n", rc_last_error( rcGpu ) ) ;
					exit( 1 ) ;
				}
				for ( i = 0 ; i < batchSize ; ++i )
				{
					strcpy( readBatch[i].seq, rcSeq.data() + rcOff[i] ) ;
					readBatch[i].correction = rcRet[i] ; readBatch[i].l = rcL[i] ; readBatch[i].m = rcM[i] ; readBatch[i].h = rcH[i] ;
					readBatch[i].badPrefix = readBatch[i].badSuffix = 0 ;   // ErrorCorrection.cpp:113-114
					if ( rcPaired )
					{
						const int j = batchSize + i ;
						strcpy( readBatch2[i].seq, rcSeq2.data() + rcOff2[i] ) ;
						readBatch2[i].correction = rcRet[j] ; readBatch2[i].l = rcL[j] ; readBatch2[i].m = rcM[j] ; readBatch2[i].h = rcH[j] ;
						readBatch2[i].badPrefix = readBatch2[i].badSuffix = 0 ;
					}
				}
			}
'''

# (first line, last line) of the reference to replace (1-based, inclusive; last < first = pure insertion
# BEFORE `first`), a token that must occur in the first replaced / following line, the new text
EDITS = [
    (16, 15, '', INCLUDES),                       # after the #include block (main.cpp:12-15)
    (193, 192, 'fpJellyFishDump', None),          # placeholder: handled below (needs argv[i + 1])
    (270, 269, 'KmerCode kcode', FORCE_BATCH_PATH),
    (294, 308, 'kmerCount', LOAD_TABLE),
    (314, 357, 'rewind', ERROR_RATE),
    (365, 364, '', RUN_PARAMS),                   # after the "Bad quality threshold" line (main.cpp:364)
    (479, 483, 'pthread_create', CORRECT_BATCH),
]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    if len(lines) < 520 or "ErrorCorrection.h" not in lines[14] or "Store kmers" not in lines[139]:
        sys.exit("apply_integration: %s is not main.cpp of Rcorrector v1.0.7" % src)
    edits = list(EDITS)
    edits[1] = (193, 192, 'fpJellyFishDump', "\t\t\trcDumpPath = argv[i + 1] ;\n")
    for first, last, token, text in sorted(edits, key=lambda e: -e[0]):
        probe = lines[first - 1] if last >= first else lines[first - 2] + lines[first - 1]
        if token and token not in probe and token not in lines[first]:
            sys.exit("apply_integration: line %d of %s does not look like v1.0.7 (no %r)" % (first, src, token))
        new = text.rstrip("\n").split("\n")
        if last >= first:
            lines[first - 1:last] = new
        else:
            lines[first - 1:first - 1] = new
    open(dst, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
