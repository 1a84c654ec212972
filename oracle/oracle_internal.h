/* oracle/oracle_internal.h — TEST INFRASTRUCTURE ONLY (see oracle.h). */
#ifndef IRS_ORACLE_INTERNAL_H
#define IRS_ORACLE_INTERNAL_H
#include "oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BLOCK 128u

typedef struct {
  const uint8_t* p;
  const uint8_t* end;
  int bad;
} orc_in;

/* Restates doc_iterator<IteratorTraits, FieldTraits> for a FREQ field with no
 * wand data.  State names follow the reference. */
typedef struct {
  orc_in in;
  int layout;
  int want_freq;
  int field_no_freq; /* FieldTraits::frequency() == false: no freq blocks, tail = vint(delta) */
  uint32_t docs[ORC_BLOCK];
  uint32_t freqs[ORC_BLOCK];
  uint32_t begin; /* index into docs; ORC_BLOCK == end */
  uint32_t left;
  uint32_t doc;  /* document attribute, 0 = invalid, UINT32_MAX = eof */
  uint32_t freq; /* frequency attribute */
  /* single_doc_iterator */
  int single;
  uint32_t next_single;
} orc_doc_iterator;


void orc_it_prepare(orc_doc_iterator* it, const uint8_t* file, uint64_t len,
                    int layout, const orc_term_meta* m, int want_freq);
/* same, for a field indexed with `wand_count` scorers (wand data in `.doc`) */
void orc_it_prepare_wand(orc_doc_iterator* it, const uint8_t* file, uint64_t len,
                         int layout, const orc_term_meta* m, int want_freq,
                         uint32_t wand_count);
/* doc_iterator::next — returns 0 at eof (doc == UINT32_MAX) */
int orc_it_next(orc_doc_iterator* it);
/* doc_iterator::seek — first doc >= target (formats_10.cpp:2304-2365; the
 * skip-list acceleration does not change the result and is not restated) */
uint32_t orc_it_seek(orc_doc_iterator* it, uint32_t target);

#ifdef __cplusplus
}
#endif
#endif
