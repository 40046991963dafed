# aha_hip.cr -- Crystal binding that re-backs Aha::AC with libaha_hip.so
# (MI355X).  Drop this file next to the reference's src/aha/ac.cr and require
# it INSTEAD of ac.cr; the public surface is unchanged:
#
#   Aha::AC.compile(keys)            (reference: src/aha/ac.cr:62-69)
#   AC#match(seq : Bytes, &block)    (src/aha/ac.cr:280-286)
#   AC#match(seq : String, &block)   (src/aha/matcher.cr:34-39)
#   AC#match(seq, sep : BitArray)    (src/aha/ac.cr:321-340, matcher.cr:41-46)
#   AC#match(seq : Array(Char))      (src/aha/ac.cr:288-295)
#   AC#match(Array(Char), sep)       (src/aha/ac.cr:342-364: the neighbour tests look at code points)
#   AC#match_longest(seq, intersectable)   (src/aha/ac.cr:297-319)
#   AC#[](Int) / AC#[](String)       (src/aha/ac.cr:41-43)
#   Aha::Hit                         (src/aha/matcher.cr:2-11, unchanged)
# plus the batch surface the reference does not have (one sequence per call there):
#   AC#match_batch(docs)             -> Array(Array(Aha::Hit)), one GPU round trip for all documents
#   Aha::ACGroup                     the same over several GPUs of one node (aha_group_*)
#
# AC.compile(da : Cedar) (src/aha/ac.cr:71) is provided for tries built by `insert` only: the keys are read back in
# id order (Cedar#[](id), cedar.cr:747-749) and compiled by the library; ids freed by Cedar#delete are rejected,
# because the library numbers keys densely (Hit#value = index in compile order).
#
# Written for the Crystal the reference pins (shard.yml: 0.23.1): integer division is `/`.
#
# UNVERIFIED: no Crystal toolchain exists in the build container or on the GPU
# box, so this file has never been compiled.  The same .so is exercised
# through the identical C ABI by the Python and C++ mirrors (aha_amd/ac.py,
# include/aha/ac.hpp).
require "bit_array"
require "./matcher"

@[Link("aha_hip")]
lib LibAhaHip
  type Ac = Void*

  struct Hit
    start : Int32
    end_ : Int32
    value : Int32
  end

  struct Options
    struct_size : UInt32
    device : Int32
    flags : UInt32
    reserved : UInt32
  end

  struct MatchParams
    struct_size : UInt32
    char_offsets : Int32
    sep_size : Int32
    sep_bits : UInt8[32]
    longest : Int32 # 0 = #match, 1 = #match_longest(intersectable: false), 2 = #match_longest(intersectable: true)
  end

  # aha_ac_info_t (ABI 8)
  struct Info
    struct_size : UInt32
    n_keys : UInt32
    n_states : UInt64
    n_slots : UInt64
    image_bytes : UInt64
    max_key_len : UInt32
    slot_bytes : UInt32
    lds_slots : UInt32
    device : Int32
    fail_s1_lo : UInt32
    fail_s2_lo : UInt32
    fail_hdr_lo : UInt32
    unit_header_beside : UInt32 # ABI 7: the traversal requests a fail header beside its probe
    unit_enabled : UInt32
    unit_slots : UInt32
    unit_syms : UInt32
    unit_multi_permille : UInt32
    unit_big_lo : UInt32
    unit_big_block : UInt32
    unit_n_low : UInt32
    unit_n_big : UInt32
    unit_base_bits : UInt32
    unit_headers : UInt32       # ABI 7: states that own a fail header
    filter_prefix_bytes : UInt32 # ABI 7: the prefix-filter engine looks at this many first bytes of a key (0: none)
    filter_words : UInt32       # ABI 7: 32-bit words of its Bloom filter
    skip_filter_words : UInt32  # ABI 8: words of the mark filter of the skip-ahead traversal (engine 6; 0: none)
    skip_pairs : UInt32         # ABI 8: two-character trie paths it holds
    pair_hash_k1 : UInt32       # ABI 8: multiplier of the second character in the pair hash
    pair_table_log2 : UInt32    # ABI 8: log2 of the pair table's 16-byte slots (0: none)
    pair_groups : UInt32        # ABI 8: its displacement bytes
    pair_engine : UInt32        # ABI 8: 1 = byte-offset matches run the pair engine (engine 7)
  end

  # aha_timing (ABI 6): filled when profiling is on
  struct Timing
    struct_size : UInt32
    n_kernels : UInt32
    ms_total : Float32
    ms_count : Float32
    ms_scan : Float32
    ms_write : Float32
    ms_aux : Float32
    n_chunks : UInt64
    n_hits : UInt64
    engine : UInt32      # 4 = character-level traversal, 2 = single-traversal engine, 1 = two-pass engine
    chunk_bytes : UInt32
    repeats : UInt32     # passes thrown away: 1 = a region overflowed and the match ran again with full-size regions
    reserved : UInt32
  end

  struct StreamSeg
    word_offset : UInt64
    n_hits : UInt64
    out_offset : UInt64
  end

  struct GroupTiming
    struct_size : UInt32
    n_devices : UInt32
    ms_match : Float32
    ms_match_max_shard : Float32
    ms_exchange : Float32
    ms_download : Float32
    n_hits : UInt64
    exchange : UInt32
    packed : UInt32
    wire_bytes : UInt64
  end

  fun aha_abi_version : UInt32
  fun aha_device_count : Int32
  fun aha_strerror(code : Int32) : UInt8*
  fun aha_last_error(ac : Ac) : UInt8*
  fun aha_ac_compile(key_bytes : UInt8*, key_offsets : UInt64*, n_keys : UInt32,
                     opts : Options*, out : Ac*, err_key : UInt32*) : Int32
  fun aha_ac_free(ac : Ac) : Void
  # a second handle for the same keys on another device: nothing is compiled again (what aha_group_compile does)
  fun aha_ac_replicate(ac : Ac, device : Int32, out : Ac*) : Int32
  fun aha_ac_info(ac : Ac, info : Info*) : Int32
  fun aha_ac_set_profiling(ac : Ac, enabled : Int32) : Int32
  fun aha_ac_last_timing(ac : Ac, t : Timing*) : Int32
  # ABI 7: the device match that also leaves the hits as the 4-byte exchange stream
  fun aha_ac_match_batch_device_stream(ac : Ac, d_corpus : UInt8*, d_doc_offsets : UInt64*, n_docs : UInt64, n_bytes : UInt64,
                                       params : MatchParams*, d_out : Hit*, cap : UInt64, d_doc_hit_offsets : UInt64*,
                                       n_hits : UInt64*, d_words : UInt32*, cap_words : UInt64, d_n_words : UInt64*,
                                       stream : Void*) : Int32
  fun aha_ac_release_scratch(ac : Ac) : Int32
  fun aha_ac_scratch_bytes(ac : Ac) : Int64
  fun aha_ac_export(ac : Ac, which : Int32, buf : Void*, cap_bytes : UInt64) : Int64
  fun aha_ac_key(ac : Ac, id : Int32, buf : UInt8*, cap : Int32) : Int32
  fun aha_ac_id(ac : Ac, key : UInt8*, len : Int32) : Int32
  fun aha_ac_save(ac : Ac, buf : Void*, cap_bytes : UInt64) : Int64
  fun aha_ac_load(buf : Void*, n_bytes : UInt64, opts : Options*, out : Ac*) : Int32
  fun aha_ac_match_bytes(ac : Ac, text : UInt8*, n : UInt64, params : MatchParams*,
                         out : Hit*, cap : UInt64, n_hits : UInt64*) : Int32
  fun aha_ac_match_batch(ac : Ac, corpus : UInt8*, doc_offsets : UInt64*, n_docs : UInt64,
                         params : MatchParams*, out : Hit*, cap : UInt64,
                         doc_hit_offsets : UInt64*, n_hits : UInt64*) : Int32

  # host corpus in, hits left on the device (d_hits: device memory of the handle's device)
  fun aha_ac_match_batch_keep(ac : Ac, corpus : UInt8*, doc_offsets : UInt64*, n_docs : UInt64,
                              params : MatchParams*, d_hits : Hit*, cap : UInt64,
                              doc_hit_offsets : UInt64*, n_hits : UInt64*) : Int32
  # exchange formats of the hit lists (all on device memory, asynchronous on `stream`)
  fun aha_ac_stream_format(ac : Ac, step_bits : UInt32*, len_bits : UInt32*) : Int32
  fun aha_ac_hits_pack_device(ac : Ac, d_hits : Hit*, n : UInt64, d_pairs : Int32*, stream : Void*) : Int32
  fun aha_ac_hits_unpack_device(ac : Ac, d_pairs : Int32*, n : UInt64, char_offsets : Int32, d_hits : Hit*,
                                stream : Void*) : Int32
  fun aha_ac_hits_pack4_device(ac : Ac, d_hits : Hit*, n : UInt64, d_words : UInt32*, cap_words : UInt64,
                               d_n_words : UInt64*, stream : Void*) : Int32
  fun aha_ac_hits_unpack4_device(ac : Ac, d_words : UInt32*, n : UInt64, char_offsets : Int32, d_hits : Hit*,
                                 stream : Void*) : Int32
  fun aha_ac_hits_unpack4_segs_device(ac : Ac, d_words : UInt32*, segs : StreamSeg*, n_segs : UInt32,
                                      char_offsets : Int32, d_hits : Hit*, stream : Void*) : Int32
  # device-resident batches: upload a corpus once, match it many times, keep the hits in HBM until they are wanted
  fun aha_ac_match_batch_device(ac : Ac, d_corpus : UInt8*, d_doc_offsets : UInt64*, n_docs : UInt64, n_bytes : UInt64,
                                params : MatchParams*, d_out : Hit*, cap : UInt64, d_doc_hit_offsets : UInt64*,
                                n_hits : UInt64*, stream : Void*) : Int32
  fun aha_buffer_alloc(device : Int32, bytes : UInt64, d_ptr : Void**) : Int32
  fun aha_buffer_free(device : Int32, d_ptr : Void*) : Int32
  fun aha_buffer_upload(device : Int32, d_dst : Void*, src : Void*, bytes : UInt64) : Int32
  fun aha_buffer_download(device : Int32, dst : Void*, d_src : Void*, bytes : UInt64) : Int32
  type Corpus = Void*
  fun aha_corpus_upload(device : Int32, corpus : UInt8*, doc_offsets : UInt64*, n_docs : UInt64, out : Corpus*) : Int32
  fun aha_corpus_free(c : Corpus) : Void
  fun aha_corpus_bytes(c : Corpus) : UInt8*
  fun aha_corpus_doc_offsets(c : Corpus) : UInt64*
  fun aha_corpus_n_docs(c : Corpus) : UInt64
  fun aha_corpus_n_bytes(c : Corpus) : UInt64
  fun aha_corpus_device(c : Corpus) : Int32

  type Group = Void*
  fun aha_group_compile(key_bytes : UInt8*, key_offsets : UInt64*, n_keys : UInt32,
                        devices : Int32*, n_devices : Int32, flags : UInt32,
                        out : Group*, err_key : UInt32*) : Int32
  fun aha_group_free(g : Group) : Void
  fun aha_group_size(g : Group) : Int32
  fun aha_group_last_error(g : Group) : UInt8*
  fun aha_group_partition(doc_offsets : UInt64*, n_docs : UInt64, n_parts : Int32, bounds : UInt64*) : Int32
  fun aha_group_download_shard(g : Group, shard : Int32, out : Hit*, cap : UInt64, n_hits : UInt64*) : Int32
  fun aha_group_last_timing(g : Group, t : GroupTiming*) : Int32
  # ABI 7: the batch resident on the group's devices
  type GroupCorpus = Void*
  fun aha_group_corpus_upload(g : Group, corpus : UInt8*, doc_offsets : UInt64*, n_docs : UInt64, out_c : GroupCorpus*) : Int32
  fun aha_group_corpus_free(c : GroupCorpus) : Void
  fun aha_group_match_batch_device(g : Group, c : GroupCorpus, params : MatchParams*, doc_hit_offsets : UInt64*,
                                   n_hits : UInt64*) : Int32
  fun aha_group_match_batch(g : Group, corpus : UInt8*, doc_offsets : UInt64*, n_docs : UInt64,
                            params : MatchParams*, out : Hit*, cap : UInt64,
                            doc_hit_offsets : UInt64*, n_hits : UInt64*) : Int32
end

module Aha
  class AC
    E_DUP_KEY  = -4
    E_CAPACITY = -6

    def initialize(@handle : LibAhaHip::Ac)
    end

    def finalize
      LibAhaHip.aha_ac_free(@handle)
    end

    def self.compile(keys : Array(String) | Array(Array(UInt8)) | Array(Bytes)) : self
      blob, offs = pack_keys(keys)
      opts = LibAhaHip::Options.new
      opts.struct_size = sizeof(LibAhaHip::Options).to_u32
      opts.device = -1
      rc = LibAhaHip.aha_ac_compile(blob.to_slice.to_unsafe, offs.to_unsafe, keys.size.to_u32,
        pointerof(opts), out handle, out bad)
      if rc == E_DUP_KEY
        raise "key:#{keys[bad]} appear twice."
      elsif rc != 0
        raise String.new(LibAhaHip.aha_strerror(rc))
      end
      new(handle)
    end

    # ACX.compile(da : Cedar) src/aha/ac.cr:71 -- see the header for the restriction
    def self.compile(da : Cedar) : self
      keys = Array(String).new
      (0...da.size).each do |id|
        keys << da[id]  # raises for an id freed by Cedar#delete
      end
      compile(keys)
    end

    protected def self.pack_keys(keys)
      blob = IO::Memory.new
      offs = Array(UInt64).new(keys.size + 1)
      offs << 0_u64
      keys.each do |k|
        bytes = k.is_a?(String) ? k.to_slice : (k.is_a?(Bytes) ? k : Slice.new(k.to_unsafe, k.size))
        blob.write bytes
        offs << blob.pos.to_u64
      end
      {blob, offs}
    end

    protected def self.params(chars : Bool, sep : BitArray?, longest : Int32 = 0) : LibAhaHip::MatchParams
      params = LibAhaHip::MatchParams.new
      params.struct_size = sizeof(LibAhaHip::MatchParams).to_u32
      params.char_offsets = chars ? 1 : 0
      params.longest = longest
      if sep
        raise "sep BitArray size > 256 is not supported" if sep.size > 256
        params.sep_size = sep.size
        bits = StaticArray(UInt8, 32).new(0_u8)
        sep.each_with_index { |b, i| bits[i >> 3] |= (1_u8 << (i & 7)) if b }
        params.sep_bits = bits
      end
      params
    end

    # New: all documents in one call (the reference is one sequence per call).  Returns the hits of every document
    # in the reference's order; `chars` selects the String overload's char offsets.
    def match_batch(docs : Array(String) | Array(Bytes), chars : Bool = false, sep : BitArray? = nil) : Array(Array(Hit))
      corpus = IO::Memory.new
      offs = Array(UInt64).new(docs.size + 1)
      offs << 0_u64
      docs.each do |d|
        corpus.write(d.is_a?(String) ? d.to_slice : d)
        offs << corpus.pos.to_u64
      end
      params = AC.params(chars, sep)
      dho = Array(UInt64).new(docs.size + 1, 0_u64)
      cap = (corpus.pos / 4 + 64).to_u64
      result = Array(Array(Hit)).new(docs.size)
      loop do
        out_buf = Pointer(LibAhaHip::Hit).malloc(cap)
        rc = LibAhaHip.aha_ac_match_batch(@handle, corpus.to_slice.to_unsafe, offs.to_unsafe, docs.size.to_u64,
          pointerof(params), out_buf, cap, dho.to_unsafe, out n)
        if rc == E_CAPACITY
          cap = n
          next
        end
        raise String.new(LibAhaHip.aha_last_error(@handle)) if rc != 0
        docs.size.times do |d|
          hits = Array(Hit).new((dho[d + 1] - dho[d]).to_i32)
          (dho[d]...dho[d + 1]).each { |i| hits << Hit.new(out_buf[i].start, out_buf[i].end_, out_buf[i].value) }
          result << hits
        end
        break
      end
      result
    end

    private def run(seq : Bytes, chars : Bool, sep : BitArray?, longest : Int32 = 0, &block)
      params = AC.params(chars, sep, longest)
      cap = (seq.size / 4 + 64).to_u64
      loop do
        out_buf = Pointer(LibAhaHip::Hit).malloc(cap)
        rc = LibAhaHip.aha_ac_match_bytes(@handle, seq.to_unsafe, seq.size.to_u64, pointerof(params),
          out_buf, cap, out n)
        if rc == E_CAPACITY
          cap = n
          next
        end
        raise String.new(LibAhaHip.aha_last_error(@handle)) if rc != 0
        n.times { |i| yield Hit.new(out_buf[i].start, out_buf[i].end_, out_buf[i].value) }
        break
      end
    end

    def match(seq : Bytes | Array(UInt8), &block)
      bytes = seq.is_a?(Bytes) ? seq : Slice.new(seq.to_unsafe, seq.size)
      run(bytes, false, nil) { |hit| yield hit }
    end

    def match(seq : String, &block)
      run(seq.to_slice, true, nil) { |hit| yield hit }
    end

    def match(seq : Array(Char) | Slice(Char), &block)
      run(String.build { |s| seq.each { |c| s << c } }.to_slice, true, nil) { |hit| yield hit }
    end

    def match(seq : Bytes | Array(UInt8), sep : BitArray, &block)
      bytes = seq.is_a?(Bytes) ? seq : Slice.new(seq.to_unsafe, seq.size)
      run(bytes, false, sep) { |hit| yield hit }
    end

    def match(seq : String, sep : BitArray, &block)
      run(seq.to_slice, true, sep) { |hit| yield hit }
    end

    # ACX#match_longest src/aha/ac.cr:297-319.  Cedar's stale END flags (cedar.cr:642-648) ARE reproduced: the library
    # replays Cedar's inserts from the keys on the first match_longest call of a handle (aha_amd/csrc/cedar_replay.cpp);
    # NUL bytes in the text behave as in the reference (value nodes, kernels.hip)
    def match_longest(seq : Bytes | Array(UInt8), intersectable = false, &block)
      bytes = seq.is_a?(Bytes) ? seq : Slice.new(seq.to_unsafe, seq.size)
      run(bytes, false, nil, intersectable ? 2 : 1) { |hit| yield hit }
    end

    def match_longest(seq : String, intersectable = false, &block)
      run(seq.to_slice, true, nil, intersectable ? 2 : 1) { |hit| yield hit }
    end

    def match_longest(seq : Array(Char) | Slice(Char), intersectable = false, &block)
      run(String.build { |s| seq.each { |c| s << c } }.to_slice, true, nil, intersectable ? 2 : 1) { |hit| yield hit }
    end

    # src/aha/ac.cr:342-364: unlike the String overload the neighbour tests look at the neighbouring CHAR's code
    # point (`chr.ord < sep.size && !sep[chr.ord]`), so they are applied here to the unfiltered byte hits.
    def match(seq : Array(Char) | Slice(Char), sep : BitArray, &block)
      raise "sep BitArray size > 256 is not supported" if sep.size > 256
      str = String.build { |s| seq.each { |c| s << c } }
      char_of_byte = Array(Int32).new(str.bytesize)
      seq.each_with_index { |c, i| c.bytesize.times { char_of_byte << i } }
      run(str.to_slice, false, nil) do |hit|
        chr_idx = char_of_byte[hit.end - 1]
        if chr_idx + 1 < seq.size
          chr = seq[chr_idx + 1]
          next if chr.ord < sep.size && !sep[chr.ord]
        end
        if hit.start > 0
          chr = seq[char_of_byte[hit.start] - 1]
          next if chr.ord < sep.size && !sep[chr.ord]
        end
        yield Hit.new(char_of_byte[hit.start], char_of_byte[hit.end - 1] + 1, hit.value)
      end
    end

    # AC#to_io / AC.from_io (src/aha/ac.cr:45-60) on the library's own container
    def to_io(io : IO, format : IO::ByteFormat = IO::ByteFormat::LittleEndian)
      n = LibAhaHip.aha_ac_save(@handle, Pointer(Void).null, 0_u64)
      raise "save failed" if n < 0
      buf = Bytes.new(n)
      LibAhaHip.aha_ac_save(@handle, buf.to_unsafe.as(Void*), n.to_u64)
      io.write buf
    end

    def self.from_io(io : IO, format : IO::ByteFormat = IO::ByteFormat::LittleEndian) : self
      data = io.gets_to_end.to_slice
      rc = LibAhaHip.aha_ac_load(data.to_unsafe.as(Void*), data.size.to_u64, Pointer(LibAhaHip::Options).null, out h)
      raise String.new(LibAhaHip.aha_strerror(rc)) if rc != 0
      new(h)
    end

    def [](sid : Int) : String
      n = LibAhaHip.aha_ac_key(@handle, sid.to_i32, Pointer(UInt8).null, 0)
      raise IndexError.new if n < 0
      buf = Bytes.new(n)
      LibAhaHip.aha_ac_key(@handle, sid.to_i32, buf.to_unsafe, n)
      String.new(buf)
    end

    def [](key : String) : Int32
      r = LibAhaHip.aha_ac_id(@handle, key.to_unsafe, key.bytesize)
      raise IndexError.new if r < 0
      r
    end
  end

  # `Aha::ACBig = ACX(Int64)` (src/aha/ac.cr:9) differs from `Aha::AC` only in the width of its node ids: what `#match` yields
  # is the same `Hit` with an Int32 value (`val.to_i32`, ac.cr:273).  The library's own numbering has no such limit below
  # 2^31 keys, so one class answers for both names.
  alias ACBig = AC

  # Several GPUs of one node behind one object: the batch is cut into contiguous, byte-balanced document ranges (one
  # per device), every device matches its range, the hit buffers are exchanged with an all-gatherv (RCCL over xGMI
  # between distinct devices) and come back in document order -- the same hits as AC#match_batch.
  class ACGroup
    def initialize(@group : LibAhaHip::Group)
    end

    def finalize
      LibAhaHip.aha_group_free(@group)
    end

    def self.compile(keys : Array(String) | Array(Bytes), devices : Array(Int32)) : self
      blob, offs = AC.pack_keys(keys)
      rc = LibAhaHip.aha_group_compile(blob.to_slice.to_unsafe, offs.to_unsafe, keys.size.to_u32,
        devices.to_unsafe, devices.size, 0_u32, out group, out bad)
      if rc == AC::E_DUP_KEY
        raise "key:#{keys[bad]} appear twice."
      elsif rc != 0
        raise String.new(LibAhaHip.aha_strerror(rc))
      end
      new(group)
    end

    def match_batch(docs : Array(String) | Array(Bytes), chars : Bool = false) : Array(Array(Hit))
      corpus = IO::Memory.new
      offs = Array(UInt64).new(docs.size + 1)
      offs << 0_u64
      docs.each do |d|
        corpus.write(d.is_a?(String) ? d.to_slice : d)
        offs << corpus.pos.to_u64
      end
      params = AC.params(chars, nil)
      dho = Array(UInt64).new(docs.size + 1, 0_u64)
      cap = (corpus.pos / 4 + 64).to_u64
      result = Array(Array(Hit)).new(docs.size)
      loop do
        out_buf = Pointer(LibAhaHip::Hit).malloc(cap)
        rc = LibAhaHip.aha_group_match_batch(@group, corpus.to_slice.to_unsafe, offs.to_unsafe, docs.size.to_u64,
          pointerof(params), out_buf, cap, dho.to_unsafe, out n)
        if rc == AC::E_CAPACITY
          cap = n
          next
        end
        raise String.new(LibAhaHip.aha_strerror(rc)) if rc != 0
        docs.size.times do |d|
          hits = Array(Hit).new((dho[d + 1] - dho[d]).to_i32)
          (dho[d]...dho[d + 1]).each { |i| hits << Hit.new(out_buf[i].start, out_buf[i].end_, out_buf[i].value) }
          result << hits
        end
        break
      end
      result
    end

    # The batch resident on the devices (aha_group_corpus_upload): every shard's documents on its device, uploaded once.
    class Corpus
      getter handle : LibAhaHip::GroupCorpus
      getter n_docs : Int32

      def initialize(@handle, @n_docs)
      end

      def finalize
        LibAhaHip.aha_group_corpus_free(@handle)
      end
    end

    def upload(docs : Array(String) | Array(Bytes)) : Corpus
      corpus = IO::Memory.new
      offs = Array(UInt64).new(docs.size + 1)
      offs << 0_u64
      docs.each do |d|
        corpus.write(d.is_a?(String) ? d.to_slice : d)
        offs << corpus.pos.to_u64
      end
      rc = LibAhaHip.aha_group_corpus_upload(@group, corpus.to_slice.to_unsafe, offs.to_unsafe, docs.size.to_u64, out c)
      raise String.new(LibAhaHip.aha_strerror(rc)) if rc != 0
      Corpus.new(c, docs.size)
    end

    # Every device matches its resident range, then the all-gatherv; the hits stay on the devices (every device holds the
    # whole ordered stream).  Returns the hit count and the per-document hit offsets; `shard_hits` reads one device's copy.
    def match_resident(c : Corpus, chars : Bool = false) : {UInt64, Array(UInt64)}
      params = AC.params(chars, nil)
      dho = Array(UInt64).new(c.n_docs + 1, 0_u64)
      rc = LibAhaHip.aha_group_match_batch_device(@group, c.handle, pointerof(params), dho.to_unsafe, out n)
      raise String.new(LibAhaHip.aha_strerror(rc)) if rc != 0
      {n, dho}
    end

    def shard_hits(shard : Int32) : Array(Hit)
      LibAhaHip.aha_group_download_shard(@group, shard, Pointer(LibAhaHip::Hit).null, 0_u64, out n)
      buf = Pointer(LibAhaHip::Hit).malloc(n + 1)
      rc = LibAhaHip.aha_group_download_shard(@group, shard, buf, n + 1, out got)
      raise String.new(LibAhaHip.aha_strerror(rc)) if rc != 0
      Array(Hit).new(got.to_i32) { |i| Hit.new(buf[i].start, buf[i].end_, buf[i].value) }
    end
  end
end
