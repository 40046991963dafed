# aha_hip.cr -- Crystal binding that re-backs Aha::AC with libaha_hip.so
# (MI355X).  Drop this file next to the reference's src/aha/ac.cr and require
# it INSTEAD of ac.cr; the public surface is unchanged:
#
#   Aha::AC.compile(keys)            (reference: src/aha/ac.cr:62-69)
#   AC#match(seq : Bytes, &block)    (src/aha/ac.cr:280-286)
#   AC#match(seq : String, &block)   (src/aha/matcher.cr:34-39)
#   AC#match(seq, sep : BitArray)    (src/aha/ac.cr:321-340, matcher.cr:41-46)
#   AC#match(seq : Array(Char))      (src/aha/ac.cr:288-295)
#   AC#[](Int) / AC#[](String)       (src/aha/ac.cr:41-43)
#   Aha::Hit                         (src/aha/matcher.cr:2-11, unchanged)
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
  end

  fun aha_strerror(code : Int32) : UInt8*
  fun aha_last_error(ac : Ac) : UInt8*
  fun aha_ac_compile(key_bytes : UInt8*, key_offsets : UInt64*, n_keys : UInt32,
                     opts : Options*, out : Ac*, err_key : UInt32*) : Int32
  fun aha_ac_free(ac : Ac) : Void
  fun aha_ac_key(ac : Ac, id : Int32, buf : UInt8*, cap : Int32) : Int32
  fun aha_ac_id(ac : Ac, key : UInt8*, len : Int32) : Int32
  fun aha_ac_save(ac : Ac, buf : Void*, cap_bytes : UInt64) : Int64
  fun aha_ac_load(buf : Void*, n_bytes : UInt64, opts : Options*, out : Ac*) : Int32
  fun aha_ac_match_bytes(ac : Ac, text : UInt8*, n : UInt64, params : MatchParams*,
                         out : Hit*, cap : UInt64, n_hits : UInt64*) : Int32
  fun aha_ac_match_batch(ac : Ac, corpus : UInt8*, doc_offsets : UInt64*, n_docs : UInt64,
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
      blob = IO::Memory.new
      offs = Array(UInt64).new(keys.size + 1)
      offs << 0_u64
      keys.each do |k|
        bytes = k.is_a?(String) ? k.to_slice : (k.is_a?(Bytes) ? k : Slice.new(k.to_unsafe, k.size))
        blob.write bytes
        offs << blob.pos.to_u64
      end
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

    private def run(seq : Bytes, chars : Bool, sep : BitArray?, &block)
      params = LibAhaHip::MatchParams.new
      params.struct_size = sizeof(LibAhaHip::MatchParams).to_u32
      params.char_offsets = chars ? 1 : 0
      if sep
        raise "sep BitArray size > 256 is not supported" if sep.size > 256
        params.sep_size = sep.size
        bits = StaticArray(UInt8, 32).new(0_u8)
        sep.each_with_index { |b, i| bits[i >> 3] |= (1_u8 << (i & 7)) if b }
        params.sep_bits = bits
      end
      cap = (seq.size // 4 + 64).to_u64
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
end
