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

/* Restates position<IteratorTraits, FieldTraits> + position_impl<..., false, false>
 * (formats_10.cpp:1457-1682): a field with POS and neither offsets nor payloads,
 * zero-based storage (formats 1_3+, one_based_position_storage() == false). */
typedef struct {
  orc_in in;
  const uint8_t* file;
  int layout;
  uint32_t pos_deltas[ORC_BLOCK];
  uint64_t pend_pos;    /* how many positions "behind" we are */
  uint64_t tail_start;  /* file pointer of the vint-coded last block */
  uint32_t tail_length;
  uint32_t buf_pos;     /* ORC_BLOCK = buffer consumed */
  uint32_t value;       /* 0 = pos_limits::invalid(), UINT32_MAX = eof */
  int one_based;        /* IteratorTraits::one_based_position_storage(): formats 1_0..1_2 */
} orc_pos_iterator;

void orc_pos_prepare(orc_pos_iterator* p, const uint8_t* pos_file, uint64_t len, int layout,
                     const orc_term_meta* m, int one_based);
void orc_pos_notify(orc_pos_iterator* p, uint32_t n); /* + clear(): doc iterator moved on */
int orc_pos_next(orc_pos_iterator* p, uint32_t freq);
uint32_t orc_pos_seek(orc_pos_iterator* p, uint32_t freq, uint32_t target);

#ifdef __cplusplus
}
#endif
#endif
